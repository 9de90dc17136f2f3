#!/bin/bash
# r9g: the dense work list for the lanes' own hits (-DSAGE_DENSE_HITS=1: the last chunk's items through select_most_intense_peak 64 at a
# time, the list in the bitmap's bytes) against the lane-by-lane walk: parity + fuzz suites on the variant, then C3 / C3T / C2 timing
OUT=gpurun_out/r9g; mkdir -p $OUT; export TMPDIR=/tmp
SAGE_HIP_LIB=$PWD/sage_amd/libsage_hip_dense.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_config_scale.py -m gpu -q -x --no-header -p no:cacheprovider > $OUT/pytest_dense.log 2>&1; tail -3 $OUT/pytest_dense.log
timeout 900 python scripts/ab_multi.py C3 --sizes 500000,m3/8 --steps 10 -- base dense base dense > $OUT/ab_C3.log 2>&1; grep RESULT -B1 $OUT/ab_C3.log
timeout 900 python scripts/ab_multi.py C3T --sizes 500000 --steps 10 -- base dense > $OUT/ab_C3T.log 2>&1; grep RESULT -B1 $OUT/ab_C3T.log
