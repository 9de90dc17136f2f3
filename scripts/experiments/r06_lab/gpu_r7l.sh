#!/bin/bash
# r7l: where rescore_kernel's time goes on the tie-rich C3T (63 % of its spectra settle a tie from the stored window counts)
OUT=gpurun_out/r7l; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python scripts/phase_clocks.py C3T 131072 > $OUT/phase_clocks_C3T.txt 2>&1; tail -4 $OUT/phase_clocks_C3T.txt
timeout 600 python scripts/phase_clocks.py C3 131072 > $OUT/phase_clocks_C3.txt 2>&1; tail -4 $OUT/phase_clocks_C3.txt
