#!/bin/bash
# round 5, call m: the wide-list path (50 and 200 PSMs per spectrum) before / after the workspace option; the whole suite
OUT=gpurun_out/r5m; mkdir -p $OUT; export TMPDIR=/tmp
AB_REPORT_PSMS=50 timeout 900 python scripts/ab_multi.py C3 --sizes 500000 --steps 4 -- prev base base:SAGE_HIP_FORCE_HUGE=1 > $OUT/c3_50psms.txt 2>&1; cat $OUT/c3_50psms.txt
AB_REPORT_PSMS=200 timeout 900 python scripts/ab_multi.py C3 --sizes 131072 --steps 3 -- prev base > $OUT/c3_200psms.txt 2>&1; cat $OUT/c3_200psms.txt
( time timeout 1200 python -m pytest tests -m gpu -q -x ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log
