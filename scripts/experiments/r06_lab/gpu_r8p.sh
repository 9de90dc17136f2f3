#!/bin/bash
OUT=gpurun_out/r8p; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 4 $OUT/pytest.log
