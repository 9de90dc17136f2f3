#!/bin/bash
# r7r: rank locate, second version (nothing of publish step 1 live across the barrier), C4 / C5 against the build before (q4)
OUT=gpurun_out/r7r; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 5 -- q4 base > $OUT/ab_C4.log 2>&1; grep RESULT -B1 $OUT/ab_C4.log
timeout 1500 python scripts/ab_multi.py C5 --sizes 200000 --steps 5 -- q4 base > $OUT/ab_C5.log 2>&1; grep RESULT -B1 $OUT/ab_C5.log
timeout 300 python scripts/tile_phase_cfg.py C4 20000 > $OUT/C4_count_phase_clocks.txt 2>&1; tail -11 $OUT/C4_count_phase_clocks.txt
