#!/bin/bash
# round 4, pass ac: the probe variant's table reads pipelined across batches (base) vs not (np)
OUT=gpurun_out/r4ac; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python scripts/ab_multi.py C3 --sizes 62500,500000 --steps 40 -- np base > $OUT/ab_C3.txt 2>&1; cat $OUT/ab_C3.txt
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_scale.py -m gpu -q -x ) > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
