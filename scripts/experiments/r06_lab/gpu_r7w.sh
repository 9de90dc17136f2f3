#!/bin/bash
# r7w: the eight shards of the mass-block plan on the final build (a rank's step at 8 GPUs), and the number of parts a small step
# is cut into (SAGE_HIP_WAYS) on one of them
OUT=gpurun_out/r7w; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python scripts/ab_multi.py C3 --sizes b0/8,b1/8,b2/8,b3/8,b4/8,b5/8,b6/8,b7/8,b0/4,b0/2,500000 --steps 30 -- base > $OUT/ab_shards.log 2>&1; grep RESULT -B1 $OUT/ab_shards.log
timeout 1500 python scripts/ab_multi.py C3 --sizes b3/8 --steps 30 -- base:SAGE_HIP_WAYS=1 base:SAGE_HIP_WAYS=2 base:SAGE_HIP_WAYS=3 base:SAGE_HIP_WAYS=4 > $OUT/ab_ways.log 2>&1; grep RESULT -B1 $OUT/ab_ways.log
