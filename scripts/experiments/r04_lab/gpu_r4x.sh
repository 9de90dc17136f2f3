#!/bin/bash
# round 4, pass x: host to host — the last piece cut in three, now that the sort is off the link
OUT=gpurun_out/r4x; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python scripts/ab_multi.py C3 --sizes 500000 --steps 30 --h2h -- base:SAGE_HIP_TAPER=0 base base:SAGE_HIP_TAPER=0 base > $OUT/ab_C3.txt 2>&1; cat $OUT/ab_C3.txt
