#!/bin/bash
# r8c: prelim_kernel at six wavefronts per SIMD without vector spills: 2 (or 3) cells per lane in flight instead of 4
OUT=gpurun_out/r8c; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python scripts/ab_multi.py C3 --sizes b3/8,500000 --steps 20 -- base w6c2 w6c2np w6c3 c3 > $OUT/ab_C3.log 2>&1; grep RESULT -B1 $OUT/ab_C3.log
