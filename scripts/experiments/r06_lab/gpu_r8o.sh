#!/bin/bash
# r8o: sorted queue + a deliberate jitter (neighbours spread by an odd multiplier inside blocks of 2^k positions), C4 with the queue kernel forced, C5
OUT=gpurun_out/r8o; mkdir -p $OUT; export TMPDIR=/tmp
F="SAGE_HIP_QUEUE_LATER=1"
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 5 -- base "base:$F" "base:$F,SAGE_HIP_DEBUG_FLAGS=$((10<<16))" "base:$F,SAGE_HIP_DEBUG_FLAGS=$((12<<16))" "base:$F,SAGE_HIP_DEBUG_FLAGS=$((13<<16))" "base:$F,SAGE_HIP_DEBUG_FLAGS=$((14<<16))" > $OUT/ab_C4.log 2>&1; grep RESULT -B1 $OUT/ab_C4.log
timeout 1500 python scripts/ab_multi.py C5 --sizes 200000 --steps 5 -- base "base:SAGE_HIP_DEBUG_FLAGS=$((10<<16))" "base:SAGE_HIP_DEBUG_FLAGS=$((13<<16))" "base:SAGE_HIP_QUEUE_LATER=0" > $OUT/ab_C5.log 2>&1; grep RESULT -B1 $OUT/ab_C5.log
