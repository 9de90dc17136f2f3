#!/bin/bash
# round 4, pass v: prelim_kernel stages the peaks behind the first window search (base) vs the commit before (h4)
OUT=gpurun_out/r4v; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python scripts/ab_multi.py C3 --sizes 62500,500000 --steps 40 -- h4 base > $OUT/ab_C3.txt 2>&1; cat $OUT/ab_C3.txt
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_scale.py -m gpu -q -x ) > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 300 python scripts/phase_clocks.py C3 131072 > $OUT/phase_C3.txt 2>&1; tail -4 $OUT/phase_C3.txt | head -1
