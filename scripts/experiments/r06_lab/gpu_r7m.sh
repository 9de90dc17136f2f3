#!/bin/bash
# r7m: rescore_kernel's in-line tie replay with the all-lanes root replacement (mask form) against the serial sift (q2), C3T and C3
OUT=gpurun_out/r7m; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "tie or equal or hyperscore" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest.log
timeout 1500 python scripts/ab_multi.py C3T --sizes 500000 --steps 10 -- q2 base > $OUT/ab_C3T.log 2>&1; grep RESULT -B1 $OUT/ab_C3T.log
timeout 1500 python scripts/ab_multi.py C3 --sizes 500000 --steps 10 -- q2 base > $OUT/ab_C3.log 2>&1; grep RESULT -B1 $OUT/ab_C3.log
