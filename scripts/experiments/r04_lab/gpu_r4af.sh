#!/bin/bash
# round 4, pass af: owners of a pass of cells by marks + prefix maximum (base) vs the binary search per cell (os0)
OUT=gpurun_out/r4af; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python scripts/ab_multi.py C3 --sizes 62500,500000 --steps 40 -- os0 base os0 base > $OUT/ab_C3.txt 2>&1; cat $OUT/ab_C3.txt
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_scale.py -m gpu -q -x ) > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
