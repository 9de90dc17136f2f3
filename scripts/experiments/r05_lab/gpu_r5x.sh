#!/bin/bash
# round 5, call x: the lightest two strides of the mass axis in 4x finer blocks (an experiment on the slowest shard), all eight shards
OUT=gpurun_out/r5x; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python scripts/ab_multi.py C3 --sizes c0/8,c1/8,c2/8,c3/8,c4/8,c5/8,c6/8,c7/8 --steps 40 -- base:AB_TIMING_EVERY=4,AB_LIGHT_REFINE=4 base:AB_TIMING_EVERY=4,AB_LIGHT_REFINE=8 base:AB_TIMING_EVERY=4 > $OUT/c3_shards.txt 2>&1; grep -E "^==|RESULT" $OUT/c3_shards.txt | cut -c1-130
