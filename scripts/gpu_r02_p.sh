#!/bin/bash
OUT=gpurun_out/r02p; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "cooperative or chimera" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -n 4 $OUT/pytest.log
scripts/ab_libs.sh C3 10 base a b c d
scripts/ab_libs.sh C5 5 base a b c d
scripts/ab_libs.sh C4 5 base d
