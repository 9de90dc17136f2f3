#!/bin/bash
OUT=gpurun_out/r02o; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_scale.py -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -n 5 $OUT/pytest.log
scripts/ab_libs.sh C4 5 base c3 c4
SAGE_HIP_NO_U8=1 scripts/ab_libs.sh C4 5 base
scripts/ab_libs.sh C5 5 base c3
SAGE_HIP_NO_U8=1 scripts/ab_libs.sh C5 5 base
scripts/ab_libs.sh C3 10 base
