#!/bin/bash
# r6r: C5's retry pass — where the two replay kernels split the queries (streams above SAGE_HIP_REPLAY_LANE_MAX words go to the
# wavefront-per-query kernel, whose offers are 2.5x cheaper since r6k); default 4096
OUT=gpurun_out/r6r; mkdir -p $OUT; export TMPDIR=/tmp
timeout 2400 python scripts/ab_multi.py C5 --sizes 200000 --steps 3 -- base "base:SAGE_HIP_REPLAY_LANE_MAX=2048" "base:SAGE_HIP_REPLAY_LANE_MAX=1024" "base:SAGE_HIP_REPLAY_LANE_MAX=512" "base:SAGE_HIP_REPLAY_LANE_MAX=256" > $OUT/ab_C5.log 2>&1
grep RESULT -B1 $OUT/ab_C5.log
