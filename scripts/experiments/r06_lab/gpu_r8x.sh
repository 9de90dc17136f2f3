#!/bin/bash
# r8x: is it the PAUSE?  The queue kernel + an idle gap of 0.5 / 1 / 2 ms in front of the count kernel (C4)
OUT=gpurun_out/r8x; mkdir -p $OUT; export TMPDIR=/tmp
F="SAGE_HIP_QUEUE_LATER=1"
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 5 -- base "base:$F" "base:$F,SAGE_HIP_QUEUE_SLEEP_US=500" "base:$F,SAGE_HIP_QUEUE_SLEEP_US=1000" "base:$F,SAGE_HIP_QUEUE_SLEEP_US=2000" > $OUT/ab_C4.log 2>&1; grep RESULT -B1 $OUT/ab_C4.log
