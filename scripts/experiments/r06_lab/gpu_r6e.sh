#!/bin/bash
# r6e: the de-spilled count kernel with one-compare index ranges (-DSAGE_HIT_RANGES=1: +20 % in round 5, when the selects spilled) at
# full size, C4 and C5, against the default and round 5.
OUT=gpurun_out/r6e; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 3 -- base hr > $OUT/ab_C4.log 2>&1; echo "ab C4 rc=$?"
grep RESULT -B1 $OUT/ab_C4.log
timeout 1500 python scripts/ab_multi.py C5 --sizes 200000 --steps 3 -- base hr r5 > $OUT/ab_C5.log 2>&1; echo "ab C5 rc=$?"
grep RESULT -B1 $OUT/ab_C5.log
