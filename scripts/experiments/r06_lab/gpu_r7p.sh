#!/bin/bash
# r7p: the count kernel's owner of a cell by rank (run-start bits per block of 64 cells + the rank of the block's first owner) instead of
# the wave walk + 6-step search, against the build before (q4); large-window parity tests first
OUT=gpurun_out/r7p; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_scale.py -q -x -k "large_window or tile or open or c4 or c5 or wide or chimera or edge" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 5 $OUT/pytest.log
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 5 -- q4 base > $OUT/ab_C4.log 2>&1; grep RESULT -B1 $OUT/ab_C4.log
timeout 1500 python scripts/ab_multi.py C5 --sizes 200000 --steps 5 -- q4 base > $OUT/ab_C5.log 2>&1; grep RESULT -B1 $OUT/ab_C5.log
