#!/bin/bash
# round 4, pass ag: small-tile size of the probe variant (2^8 .. 2^11 peptides per tile)
OUT=gpurun_out/r4ag; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python scripts/ab_multi.py C3 --sizes 62500,500000 --steps 40 -- base:SAGE_HIP_TILE2_SHIFT=11 base:SAGE_HIP_TILE2_SHIFT=10 base:SAGE_HIP_TILE2_SHIFT=9 base:SAGE_HIP_TILE2_SHIFT=8 > $OUT/ab_C3b.txt 2>&1; cat $OUT/ab_C3b.txt
