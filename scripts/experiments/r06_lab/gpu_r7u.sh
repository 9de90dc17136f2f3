#!/bin/bash
# r7u: do the tests reach the re-mark path?  The small-marks build with the re-mark left empty must FAIL the large-window tests.
OUT=gpurun_out/r7u; mkdir -p $OUT; export TMPDIR=/tmp
SAGE_HIP_LIB=$PWD/sage_amd/libsage_hip_mbx.so timeout 1200 python -m pytest tests/test_gpu_parity.py -q -k "large_window" > $OUT/pytest_mbx.log 2>&1; echo "pytest (re-mark broken) rc=$?"; tail -n 8 $OUT/pytest_mbx.log
