#!/bin/bash
# round 5, call g: 16 snake-ordered mass blocks per rank, kernel events on every 4th step; the suite on the new build
OUT=gpurun_out/r5g; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q -x ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
timeout 1200 python scripts/ab_multi.py C3 --sizes c0/8,c1/8,c2/8,c3/8,c4/8,c5/8,c6/8,c7/8,c0/4,c3/4,c0/2,500000 --steps 40 -- base:AB_TIMING_EVERY=4 base:AB_TIMING_EVERY=4,AB_BLOCKS=8 > $OUT/c3_shards.txt 2>&1; cat $OUT/c3_shards.txt
