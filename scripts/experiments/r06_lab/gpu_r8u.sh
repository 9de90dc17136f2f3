#!/bin/bash
# r8u: the order-free select and the (few, serial) replays of the first pass side by side on two streams: tests, C4 / C5 against q8
OUT=gpurun_out/r8u; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_scale.py -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest.log
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 5 -- q8 base > $OUT/ab_C4.log 2>&1; grep RESULT -B1 $OUT/ab_C4.log
timeout 1500 python scripts/ab_multi.py C5 --sizes 200000 --steps 5 -- q8 base > $OUT/ab_C5.log 2>&1; grep RESULT -B1 $OUT/ab_C5.log
