#!/bin/bash
# r6j: with 40 % less line traffic in prelim_kernel (succinct table), the occupancy / windows-in-flight knobs again:
# 6 wavefronts per SIMD (80 VGPRs), 4 windows per lane in flight (one batch covers a C3 spectrum), the same at 4 wavefronts per SIMD.
OUT=gpurun_out/r6j; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python scripts/ab_multi.py C3 --sizes 62500,500000 --steps 20 -- base w6 ppl4 ppl4w4 > $OUT/ab_C3.log 2>&1; echo "ab rc=$?"
grep RESULT -B1 $OUT/ab_C3.log
