#!/bin/bash
# round 4, pass aa: the small step after the streaming-pipeline changes (stream -> hardware queue mapping)
OUT=gpurun_out/r4aa; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python scripts/ab_multi.py C3 --sizes 62500,500000 --steps 40 --h2h -- base base > $OUT/ab_C3.txt 2>&1; cat $OUT/ab_C3.txt
