#!/bin/bash
# r8w: the arrival order of the atomics is a near-random shuffle inside ~14 000 schedule positions (gpu_r8v.sh) — the queue kernel with a
# pseudo-random bijection inside blocks of 2^12 .. 2^16 positions, C4 (queue kernel forced) and C5
OUT=gpurun_out/r8w; mkdir -p $OUT; export TMPDIR=/tmp
F="SAGE_HIP_QUEUE_LATER=1"
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 5 -- base "base:$F,SAGE_HIP_DEBUG_FLAGS=$((12<<16))" "base:$F,SAGE_HIP_DEBUG_FLAGS=$((13<<16))" "base:$F,SAGE_HIP_DEBUG_FLAGS=$((14<<16))" "base:$F,SAGE_HIP_DEBUG_FLAGS=$((15<<16))" "base:$F,SAGE_HIP_DEBUG_FLAGS=$((16<<16))" > $OUT/ab_C4.log 2>&1; grep RESULT -B1 $OUT/ab_C4.log
timeout 1500 python scripts/ab_multi.py C5 --sizes 200000 --steps 5 -- base "base:SAGE_HIP_DEBUG_FLAGS=$((13<<16))" "base:SAGE_HIP_DEBUG_FLAGS=$((15<<16))" > $OUT/ab_C5.log 2>&1; grep RESULT -B1 $OUT/ab_C5.log
