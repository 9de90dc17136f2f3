#!/bin/bash
# round 4, pass ai: sort keys in the bitmap's bytes (base); a 32 768-bin bitmap at 1/16 Da (bm15); the full GPU suite
OUT=gpurun_out/r4ai; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python scripts/ab_multi.py C3 --sizes 500000 --steps 40 -- base bm15 base bm15 > $OUT/ab_C3.txt 2>&1; cat $OUT/ab_C3.txt
( timeout 1200 python -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
