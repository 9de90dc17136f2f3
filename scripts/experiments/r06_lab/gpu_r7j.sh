#!/bin/bash
# r7j: where the two replay kernels split the queries now that the wavefront kernel's offers are cheap
OUT=gpurun_out/r7j; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python scripts/ab_multi.py C5 --sizes 200000 --steps 5 -- "base:SAGE_HIP_REPLAY_LANE_MAX=512" "base:SAGE_HIP_REPLAY_LANE_MAX=384" "base:SAGE_HIP_REPLAY_LANE_MAX=256" "base:SAGE_HIP_REPLAY_LANE_MAX=128" "base:SAGE_HIP_REPLAY_LANE_MAX=64" "base:SAGE_HIP_REPLAY_WAVE_MAX=100000000" > $OUT/ab_C5.log 2>&1; grep RESULT -B1 $OUT/ab_C5.log
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 5 -- "base:SAGE_HIP_REPLAY_LANE_MAX=256" "base:SAGE_HIP_REPLAY_WAVE_MAX=100000000" > $OUT/ab_C4.log 2>&1; grep RESULT -B1 $OUT/ab_C4.log
