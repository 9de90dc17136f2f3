#!/bin/bash
# r8n: is it the ORDER?  The atomics' arrival order under other launch schedules of prelim_kernel (C4: XCD chunk 0 / 64 / 8192, heaviest first)
OUT=gpurun_out/r8n; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 5 -- base base:SAGE_HIP_XCD_CHUNK=0 base:SAGE_HIP_XCD_CHUNK=64 base:SAGE_HIP_XCD_CHUNK=8192 base:SAGE_HIP_SCHED_DESC=1 > $OUT/ab_C4.log 2>&1; grep RESULT -B1 $OUT/ab_C4.log
