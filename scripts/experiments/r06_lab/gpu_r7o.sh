#!/bin/bash
# r7o: prelim_kernel hands a spectrum whose first window is beyond the LDS counters to the large-window kernels before staging its peaks
OUT=gpurun_out/r7o; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest.log
timeout 1500 python scripts/ab_multi.py C5 --sizes 200000 --steps 5 -- q3 base > $OUT/ab_C5.log 2>&1; grep RESULT -B1 $OUT/ab_C5.log
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 5 -- q3 base > $OUT/ab_C4.log 2>&1; grep RESULT -B1 $OUT/ab_C4.log
timeout 1500 python scripts/ab_multi.py C3 --sizes 500000 --steps 10 -- q3 base > $OUT/ab_C3.log 2>&1; grep RESULT -B1 $OUT/ab_C3.log
