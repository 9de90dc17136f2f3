#!/bin/bash
# r7v: final form of the rank locate (re-mark reachable through SAGE_HIP_DEBUG_FLAGS=2048): parity + config-scale suites, C4 / C5 / C3 against q4
OUT=gpurun_out/r7v; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_scale.py -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest.log
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 5 -- q4 base > $OUT/ab_C4.log 2>&1; grep RESULT -B1 $OUT/ab_C4.log
timeout 1500 python scripts/ab_multi.py C5 --sizes 200000 --steps 5 -- q4 base > $OUT/ab_C5.log 2>&1; grep RESULT -B1 $OUT/ab_C5.log
