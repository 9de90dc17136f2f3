#!/bin/bash
# r8r: 3 / 4 / 6 fetches of a candidate stream in flight
OUT=gpurun_out/r8r; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python scripts/ab_multi.py C5 --sizes 200000 --steps 5 -- q7 base d4 d6 > $OUT/ab_C5.log 2>&1; grep RESULT -B1 $OUT/ab_C5.log
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 5 -- base d6 > $OUT/ab_C4.log 2>&1; grep RESULT -B1 $OUT/ab_C4.log
