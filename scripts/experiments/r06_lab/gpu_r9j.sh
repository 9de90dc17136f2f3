#!/bin/bash
# r9j: dense work list, phase C over the lane's MATCHED items only (-DSAGE_DENSE_HITS=2: a word of flags at a time) against item by item
OUT=gpurun_out/r9j; mkdir -p $OUT; export TMPDIR=/tmp
SAGE_HIP_LIB=$PWD/sage_amd/libsage_hip_dense2.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x --no-header -p no:cacheprovider > $OUT/pytest_dense2.log 2>&1; tail -2 $OUT/pytest_dense2.log
timeout 900 python scripts/ab_multi.py C3 --sizes 500000 --steps 10 -- base dense dense2 base dense dense2 base dense dense2 > $OUT/ab_C3.log 2>&1; grep RESULT -B1 $OUT/ab_C3.log | cut -c1-120
timeout 900 python scripts/ab_multi.py C3T --sizes 500000 --steps 10 -- base dense dense2 > $OUT/ab_C3T.log 2>&1; grep RESULT -B1 $OUT/ab_C3T.log | cut -c1-120
