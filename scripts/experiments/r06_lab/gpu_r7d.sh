#!/bin/bash
# r7d: the 256-window batch again with LDS to spare (SAGE_HIP_WCAP=256 shrinks the counters: r7c's p4np ran at 4 wavefronts per SIMD)
OUT=gpurun_out/r7d; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python scripts/ab_multi.py C3 --sizes 500000 --steps 20 -- base base:SAGE_HIP_WCAP=256 p4np:SAGE_HIP_WCAP=256 p3np:SAGE_HIP_WCAP=256 w6:SAGE_HIP_WCAP=256 > $OUT/ab_C3.log 2>&1; grep RESULT -B1 $OUT/ab_C3.log
