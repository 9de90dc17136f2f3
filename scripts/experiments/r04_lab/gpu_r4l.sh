#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4l; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_scale.py -m gpu -q -x -k "equal_hyperscores or platform_libm or narrow_search or c3t or c3_" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
timeout 900 python scripts/ab_multi.py C3 --sizes 62500,125000,250000,500000 --steps 30 -- tk base > $OUT/ab_C3.txt 2>&1; echo "ab rc=$?"
cat $OUT/ab_C3.txt
timeout 900 python scripts/ab_multi.py C3T --sizes 62500,500000 --steps 10 -- tk base > $OUT/ab_C3T.txt 2>&1; echo "ab rc=$?"
cat $OUT/ab_C3T.txt
