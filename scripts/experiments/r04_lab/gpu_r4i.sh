#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4i; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -8 $OUT/pytest_gpu.log
timeout 900 python scripts/ab_multi.py C3 --sizes 62500,500000 --steps 30 -- p1 base > $OUT/ab_C3.txt 2>&1; echo "ab rc=$?"
cat $OUT/ab_C3.txt
