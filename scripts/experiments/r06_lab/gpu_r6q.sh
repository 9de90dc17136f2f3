#!/bin/bash
# r6q: the balanced scan with / without round 5's forms kept for windows that cover the tile (-DSAGE_TILE_WHOLE_FAST=0), C4 and C5
OUT=gpurun_out/r6q; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 3 -- base gen > $OUT/ab_C4.log 2>&1; grep RESULT -B1 $OUT/ab_C4.log
timeout 1500 python scripts/ab_multi.py C5 --sizes 200000 --steps 3 -- base gen > $OUT/ab_C5.log 2>&1; grep RESULT -B1 $OUT/ab_C5.log
