#!/bin/bash
# round 5, call r: count kernel variants (linear owner walk over running totals, with / without the one-compare ranges) against the build before
OUT=gpurun_out/r5r; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python scripts/ab_multi.py C4 --sizes 100000 --steps 6 -- prev lin1 lin0 > $OUT/c4.txt 2>&1; cat $OUT/c4.txt
timeout 900 python scripts/ab_multi.py C5 --sizes 200000 --steps 6 -- prev lin1 lin0 > $OUT/c5.txt 2>&1; cat $OUT/c5.txt
