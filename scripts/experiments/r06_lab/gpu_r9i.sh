#!/bin/bash
# r9i: dense work list: cooperative matching from 4 / 6 / 8 hits, up to 4 heavy candidates, none at all (flag 32), every heavy one (flag 64)
OUT=gpurun_out/r9i; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python scripts/ab_multi.py C3 --sizes 500000 --steps 10 -- base dense dmh8 dmh6 dmh4 dmh8l4 dense:SAGE_HIP_DEBUG_FLAGS=32 dmh8:SAGE_HIP_DEBUG_FLAGS=64 dmh8 base > $OUT/ab_C3.log 2>&1; grep RESULT -B1 $OUT/ab_C3.log
timeout 900 python scripts/ab_multi.py C3T --sizes 500000 --steps 10 -- base dmh8 dmh6 dmh4 dmh8l4 dense:SAGE_HIP_DEBUG_FLAGS=32 > $OUT/ab_C3T.log 2>&1; grep RESULT -B1 $OUT/ab_C3T.log
