#!/bin/bash
# r6s: the lane-per-query replay skims each lane's stream to its next entering offer before the wavefront sifts.  Exact-path tests;
# C5 at full size, with the split at 2048 (default) and 4096 / 8192 (more queries for the lane kernel)
OUT=gpurun_out/r6s; mkdir -p $OUT; export TMPDIR=/tmp
SAGE_HIP_REPLAY_WAVE_MAX=0 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "tile or open or wide or chimera or large or exact or equal" > $OUT/pytest_lane.log 2>&1; echo "pytest (lane kernel forced) rc=$?"; tail -n 3 $OUT/pytest_lane.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_scale.py -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest.log
timeout 2400 python scripts/ab_multi.py C5 --sizes 200000 --steps 3 -- base "base:SAGE_HIP_REPLAY_LANE_MAX=4096" "base:SAGE_HIP_REPLAY_LANE_MAX=8192" "base:SAGE_HIP_REPLAY_LANE_MAX=1024" > $OUT/ab_C5.log 2>&1
grep RESULT -B1 $OUT/ab_C5.log
