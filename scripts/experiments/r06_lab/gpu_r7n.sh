#!/bin/bash
OUT=gpurun_out/r7n; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python scripts/phase_clocks.py C3T 131072 > $OUT/phase_clocks_C3T.txt 2>&1; tail -4 $OUT/phase_clocks_C3T.txt
