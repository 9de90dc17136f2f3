#!/bin/bash
# r6t: readers of a query's candidate directory find a word's run by marks + a DPP prefix maximum instead of a six-step search over
# ds_bpermute.  Parity files; C5 / C4 at full size.
OUT=gpurun_out/r6t; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_scale.py -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest.log
timeout 1500 python scripts/ab_multi.py C5 --sizes 200000 --steps 3 -- base > $OUT/ab_C5.log 2>&1; grep RESULT -B1 $OUT/ab_C5.log
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 3 -- base > $OUT/ab_C4.log 2>&1; grep RESULT -B1 $OUT/ab_C4.log
