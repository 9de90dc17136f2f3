#!/bin/bash
# round 4, pass ah: cells per Da of the small tiles' position table (16 / 32 / 64 / 128)
OUT=gpurun_out/r4ah; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python scripts/ab_multi.py C3 --sizes 62500,500000 --steps 40 -- base base:SAGE_HIP_LUT2_SCALE=64 base:SAGE_HIP_LUT2_SCALE=128 base:SAGE_HIP_LUT2_SCALE=16 > $OUT/ab_C3.txt 2>&1; cat $OUT/ab_C3.txt
