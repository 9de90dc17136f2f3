#!/bin/bash
# the short divisions as an INSTANCE of rescore_kernel (no run-time choice inside the kernel): the division test, C3 with and without
OUT=gpurun_out/r6b; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q -x -k "divisions or narrow_search_known or report_psms" ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
timeout 1200 python scripts/ab_multi.py C3 --sizes c0/8,500000 --steps 40 -- base:AB_TIMING_EVERY=4 base:AB_TIMING_EVERY=4,SAGE_HIP_TOL_MODE_MASK=1 base:AB_TIMING_EVERY=4 base:AB_TIMING_EVERY=4,SAGE_HIP_TOL_MODE_MASK=1 > $OUT/c3_div.txt 2>&1; cat $OUT/c3_div.txt
