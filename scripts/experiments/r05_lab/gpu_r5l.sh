#!/bin/bash
# round 5, call l: wide-list kernels with their lists in a global-memory workspace (report_psms 1100, eleven isotope errors x 500 PSMs,
# the workspace forced on small lists); C3 with 50 PSMs per spectrum before / after (the LDS arrays are sized by the list now)
OUT=gpurun_out/r5l; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "report_psms_beyond" ) > $OUT/pytest_bigk.log 2>&1; echo "pytest bigk rc=$?"; tail -30 $OUT/pytest_bigk.log
AB_REPORT_PSMS=50 timeout 600 python scripts/ab_multi.py C3 --sizes 131072 --steps 5 -- base > $OUT/c3_50psms.txt 2>&1; cat $OUT/c3_50psms.txt
