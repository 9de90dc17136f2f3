#!/bin/bash
# round 5, call u: host-to-host rate of C3 (sage_hip_score_batch) by chunk size on the round-5 build
OUT=gpurun_out/r5u; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python scripts/ab_multi.py C3 --sizes 500000 --steps 12 --h2h -- base base:SAGE_HIP_CHUNK=65536 base:SAGE_HIP_CHUNK=98304 base:SAGE_HIP_CHUNK=163840 base:SAGE_HIP_CHUNK=262144 base:SAGE_HIP_TIMING=1 > $OUT/c3_h2h.txt 2>&1; grep -E "^==|RESULT" $OUT/c3_h2h.txt | cut -c1-260
