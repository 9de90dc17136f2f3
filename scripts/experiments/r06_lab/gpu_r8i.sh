#!/bin/bash
# r8i: are the count kernel's workgroups in lockstep behind a sorted queue?  The three workgroups of a compute unit started 2 048 / 6 400 cycles apart
OUT=gpurun_out/r8i; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 5 -- q5 base st32 st100 > $OUT/ab_C4.log 2>&1; grep RESULT -B1 $OUT/ab_C4.log
