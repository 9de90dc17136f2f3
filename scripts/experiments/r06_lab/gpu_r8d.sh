#!/bin/bash
# r8d: why does prelim_kernel take 2.3 ms on C5, where it only finds every spectrum's window too large?  Its phase clocks.
OUT=gpurun_out/r8d; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python scripts/phase_clocks.py C5 40000 > $OUT/phase_clocks_C5.txt 2>&1; tail -5 $OUT/phase_clocks_C5.txt
