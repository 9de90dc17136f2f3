#!/bin/bash
# round 5, call f: snake order of the mass blocks (all eight shards), what the step-timing events cost, phase clocks of the count kernel
OUT=gpurun_out/r5f; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python scripts/ab_multi.py C3 --sizes c0/8,c1/8,c2/8,c3/8,c4/8,c5/8,c6/8,c7/8,500000 --steps 40 -- base > $OUT/c3_shards.txt 2>&1; cat $OUT/c3_shards.txt
timeout 600 python scripts/ab_multi.py C3 --sizes c0/8,c7/8,500000 --steps 40 -- base:SAGE_HIP_NO_STEP_TIMING=1 > $OUT/c3_notiming.txt 2>&1; cat $OUT/c3_notiming.txt
timeout 300 python scripts/tile_probe.py 4000 open > $OUT/tile_probe_open.txt 2>&1; tail -4 $OUT/tile_probe_open.txt
timeout 300 python scripts/tile_probe.py 4000 wide > $OUT/tile_probe_wide.txt 2>&1; tail -4 $OUT/tile_probe_wide.txt
