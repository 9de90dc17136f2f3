#!/bin/bash
# r7t: the count kernel's re-mark path: a build whose run-start marks cover exactly one round of cells (-DSAGE_MARK_BLOCKS8=24
# -DSAGE_MARK_BLOCKS16=32), every larger unit re-marks; the large-window tests (with the new ±0.5 Da fragment case) on it and on the default build
OUT=gpurun_out/r7t; mkdir -p $OUT; export TMPDIR=/tmp
SAGE_HIP_LIB=$PWD/sage_amd/libsage_hip_mb.so timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "large_window or tile or open or wide or chimera or edge or report_psms" > $OUT/pytest_mb.log 2>&1; echo "pytest (24 / 32 mark blocks) rc=$?"; tail -n 3 $OUT/pytest_mb.log
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "large_window or tile or open or wide or chimera or edge or report_psms" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest.log
SAGE_HIP_LIB=$PWD/sage_amd/libsage_hip_mb.so timeout 600 python scripts/tile_phase_cfg.py C4 5000 > $OUT/C4_mb.txt 2>&1; tail -3 $OUT/C4_mb.txt
