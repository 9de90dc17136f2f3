#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4k; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -5 $OUT/pytest_gpu.log
timeout 900 python scripts/ab_multi.py C4 --sizes 100000 --steps 5 -- p1 base > $OUT/ab_C4.txt 2>&1; echo "ab rc=$?"
cat $OUT/ab_C4.txt
timeout 900 python scripts/ab_multi.py C5 --sizes 200000 --steps 5 -- p1 base > $OUT/ab_C5.txt 2>&1; echo "ab rc=$?"
cat $OUT/ab_C5.txt
