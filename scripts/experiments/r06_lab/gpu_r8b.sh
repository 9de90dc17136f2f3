#!/bin/bash
# r8b: parts per step at full size on the round's kernels and schedule (SAGE_HIP_WAYS 1 / 2 / 4 at 500 000 and 253 000 spectra)
OUT=gpurun_out/r8b; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python scripts/ab_multi.py C3 --sizes b0/2,500000 --steps 20 -- base:SAGE_HIP_WAYS=1 base:SAGE_HIP_WAYS=2 base:SAGE_HIP_WAYS=4 > $OUT/ab_C3.log 2>&1; grep RESULT -B1 $OUT/ab_C3.log
