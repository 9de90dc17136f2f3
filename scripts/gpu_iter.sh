#!/bin/bash
# one iteration on the GPU box: the parity suites, then quick bench lines.  usage: gpurun -- scripts/gpu_iter.sh [configs...]
OUT=gpurun_out/iter; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_scale.py -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -n 4 $OUT/pytest.log
for C in ${@:-C4 C5 C3}; do scripts/ab_libs.sh $C 5 base; done
