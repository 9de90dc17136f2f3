#!/bin/bash
# r7x: a rank's shard at 8 GPUs again: heaviest precursors first (SAGE_HIP_SCHED_DESC), the XCD chunk of the schedule, two parts at 4 GPUs' shard size
OUT=gpurun_out/r7x; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python scripts/ab_multi.py C3 --sizes b3/8 --steps 30 -- base base:SAGE_HIP_SCHED_DESC=1 base:SAGE_HIP_XCD_CHUNK=256 base:SAGE_HIP_XCD_CHUNK=4096 > $OUT/ab_b38.log 2>&1; grep RESULT -B1 $OUT/ab_b38.log
timeout 1500 python scripts/ab_multi.py C3 --sizes b0/4 --steps 30 -- base base:SAGE_HIP_WAYS=2 > $OUT/ab_b04.log 2>&1; grep RESULT -B1 $OUT/ab_b04.log
