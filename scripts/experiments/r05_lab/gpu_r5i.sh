#!/bin/bash
# round 5, call i: the exact retry pass launched only when the step before had retries (suite incl. both routes on every scorer);
# the shards again; the host's clock of a small step
OUT=gpurun_out/r5i; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q -x ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest_gpu.log
timeout 1200 python scripts/ab_multi.py C3 --sizes c0/8,c1/8,c2/8,c3/8,c4/8,c5/8,c6/8,c7/8,c0/4,c0/2,500000 --steps 40 -- base:AB_TIMING_EVERY=4 > $OUT/c3_shards.txt 2>&1; cat $OUT/c3_shards.txt
SAGE_HIP_STEP_TRACE=1 timeout 600 python bench.py --config C3 --slice 0/8 --steps 12 --warmup 3 --no-cpu-baseline --no-traffic --no-extras 2> $OUT/step_trace.txt | cut -c1-300; tail -8 $OUT/step_trace.txt
timeout 600 python scripts/ab_multi.py C3T --sizes c0/8,500000 --steps 20 -- base:AB_TIMING_EVERY=4 > $OUT/c3t.txt 2>&1; cat $OUT/c3t.txt
