#!/bin/bash
# r8f: the order of the large-window queue: prelim_kernel's schedule order (XCD-chunked), plain ascending mass (4096), descending (8192)
OUT=gpurun_out/r8f; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 5 -- q5 base base:SAGE_HIP_DEBUG_FLAGS=4096 base:SAGE_HIP_DEBUG_FLAGS=8192 > $OUT/ab_C4.log 2>&1; grep RESULT -B1 $OUT/ab_C4.log
timeout 1500 python scripts/ab_multi.py C5 --sizes 200000 --steps 5 -- q5 base base:SAGE_HIP_DEBUG_FLAGS=4096 base:SAGE_HIP_DEBUG_FLAGS=8192 > $OUT/ab_C5.log 2>&1; grep RESULT -B1 $OUT/ab_C5.log
