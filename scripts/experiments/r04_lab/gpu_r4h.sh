#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4h; mkdir -p $OUT
timeout 900 python scripts/ab_multi.py C3 --sizes 500000 --steps 30 -- p1 base oldlog notie > $OUT/ab_C3.txt 2>&1; echo "ab rc=$?"
cat $OUT/ab_C3.txt
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -15 $OUT/pytest_gpu.log
