#!/bin/bash
# r9c: the queue kernel on C4 once more, on the count kernel with 2 cells per thread
OUT=gpurun_out/r9c; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 4 -- base base:SAGE_HIP_QUEUE_LATER=1 > $OUT/ab_C4.log 2>&1; grep RESULT -B1 $OUT/ab_C4.log
