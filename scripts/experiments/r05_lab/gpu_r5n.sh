#!/bin/bash
# round 5, call n: the workspace as instances of their own (LDS instances keep LDS pointers): 50 / 200 PSMs per spectrum again + the wide-list tests
OUT=gpurun_out/r5n; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "report_psms or beyond_1023 or five_thousand" ) > $OUT/pytest_bigk.log 2>&1; echo "pytest bigk rc=$?"; tail -5 $OUT/pytest_bigk.log
AB_REPORT_PSMS=50 timeout 900 python scripts/ab_multi.py C3 --sizes 500000 --steps 4 -- prev base base:SAGE_HIP_FORCE_HUGE=1 > $OUT/c3_50psms.txt 2>&1; cat $OUT/c3_50psms.txt
AB_REPORT_PSMS=200 timeout 900 python scripts/ab_multi.py C3 --sizes 131072 --steps 3 -- prev base > $OUT/c3_200psms.txt 2>&1; cat $OUT/c3_200psms.txt
