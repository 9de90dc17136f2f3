#!/bin/bash
# r6w: count kernel — a wavefront skips a cell slot none of its lanes holds a cell in (the branch-free entry tests, idle atomics and
# bookkeeping ran regardless).  Parity files; C4 / C5 at full size.
OUT=gpurun_out/r6w; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_scale.py -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest.log
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 3 -- base > $OUT/ab_C4.log 2>&1; grep RESULT -B1 $OUT/ab_C4.log
timeout 1500 python scripts/ab_multi.py C5 --sizes 200000 --steps 3 -- base > $OUT/ab_C5.log 2>&1; grep RESULT -B1 $OUT/ab_C5.log
