#!/bin/bash
# round 5, call t: heaviest-first schedule inside the parts of a multi-part step (default now); suite; the eight shards
OUT=gpurun_out/r5t; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q -x ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
timeout 1200 python scripts/ab_multi.py C3 --sizes c0/8,c1/8,c2/8,c3/8,c4/8,c5/8,c6/8,c7/8,500000 --steps 40 -- base:AB_TIMING_EVERY=4 > $OUT/c3_shards.txt 2>&1; cat $OUT/c3_shards.txt
timeout 600 python scripts/ab_multi.py C2 --sizes 50000 --steps 40 -- base:AB_TIMING_EVERY=4 > $OUT/c2.txt 2>&1; cat $OUT/c2.txt
