#!/bin/bash
# r8k: does the statistics counter of the in-line tie settlement (one atomic per tied spectrum on one address: 312 816 per C3T step) cost anything?
OUT=gpurun_out/r8k; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python scripts/ab_multi.py C3T --sizes 500000 --steps 10 -- base nts > $OUT/ab_C3T.log 2>&1; grep RESULT -B1 $OUT/ab_C3T.log
