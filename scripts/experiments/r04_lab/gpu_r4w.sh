#!/bin/bash
# round 4, pass w: the build-time knobs again on the current kernels (windows per lane, cells in flight, waves per SIMD)
OUT=gpurun_out/r4w; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python scripts/ab_multi.py C3 --sizes 500000 --steps 40 -- base pl4 pc2 pc8 w6 rw6 > $OUT/ab_C3.txt 2>&1; cat $OUT/ab_C3.txt
