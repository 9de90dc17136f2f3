#!/bin/bash
# round 5, call k: spectra of 5 400 peaks (windows in global memory / raised LDS limits); the whole suite on the build
OUT=gpurun_out/r5k; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "five_thousand or thousand_peaks" ) > $OUT/pytest_big.log 2>&1; echo "pytest big rc=$?"; tail -25 $OUT/pytest_big.log
( time timeout 1200 python -m pytest tests -m gpu -q -x ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest_gpu.log
