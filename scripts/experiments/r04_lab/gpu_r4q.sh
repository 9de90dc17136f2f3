#!/bin/bash
# round 4, pass q: charge-specialised division + unspilled zeros (base) and the cooperative-lookup thresholds
OUT=gpurun_out/r4q; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python scripts/ab_multi.py C3 --sizes 500000 --steps 40 -- h0 base c12 c8 c8l4 c6l3 > $OUT/ab_C3.txt 2>&1; cat $OUT/ab_C3.txt
timeout 300 python scripts/phase_clocks.py C3 131072 > $OUT/phase_C3.txt 2>&1; tail -2 $OUT/phase_C3.txt | head -1
