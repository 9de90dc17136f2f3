#!/bin/bash
# r8m: the three workgroups of a compute unit started 1 536 / 3 072 / 4 096 cycles apart, on C5 (sorted queue) and C4 (arrival order)
OUT=gpurun_out/r8m; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python scripts/ab_multi.py C5 --sizes 200000 --steps 5 -- base st24 st48 st64 > $OUT/ab_C5.log 2>&1; grep RESULT -B1 $OUT/ab_C5.log
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 5 -- base st24 st48 st64 > $OUT/ab_C4.log 2>&1; grep RESULT -B1 $OUT/ab_C4.log
