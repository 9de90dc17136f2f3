#!/bin/bash
# r7i: the wavefront replay's path from per-lane ancestor masks (two ANDs and a compare per offer) instead of the scalar loop, against
# the build before (q1); large-window parity tests first
OUT=gpurun_out/r7i; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_scale.py -q -x -k "large_window or tile or open or c4 or c5 or wide or chimera" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest.log
timeout 1500 python scripts/ab_multi.py C5 --sizes 200000 --steps 5 -- q1 base "base:SAGE_HIP_REPLAY_LANE_MAX=1024" "base:SAGE_HIP_REPLAY_LANE_MAX=512" > $OUT/ab_C5.log 2>&1; grep RESULT -B1 $OUT/ab_C5.log
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 5 -- q1 base "base:SAGE_HIP_REPLAY_LANE_MAX=1024" > $OUT/ab_C4.log 2>&1; grep RESULT -B1 $OUT/ab_C4.log
