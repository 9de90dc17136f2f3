#!/bin/bash
# r8g: the queue's order again: neighbours shuffled inside blocks of 4096 (16384), 1024 interleaved sub-sequences (32768), on top of the schedule order / of plain ascending (4096)
OUT=gpurun_out/r8g; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 5 -- q5 base:SAGE_HIP_DEBUG_FLAGS=16384 base:SAGE_HIP_DEBUG_FLAGS=20480 base:SAGE_HIP_DEBUG_FLAGS=32768 base:SAGE_HIP_DEBUG_FLAGS=36864 > $OUT/ab_C4.log 2>&1; grep RESULT -B1 $OUT/ab_C4.log
timeout 1500 python scripts/ab_multi.py C5 --sizes 200000 --steps 5 -- base:SAGE_HIP_DEBUG_FLAGS=4096 base:SAGE_HIP_DEBUG_FLAGS=20480 base:SAGE_HIP_DEBUG_FLAGS=36864 > $OUT/ab_C5.log 2>&1; grep RESULT -B1 $OUT/ab_C5.log
