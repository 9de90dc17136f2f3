#!/bin/bash
# r9p: the short-division instance (rescore_kernel<false, false, true>, SAGE_HIP_SHORT_DIVISIONS=1) re-measured on the round's kernel
# (round 5: 1-2 % slower — a register-allocation effect of the kernel as it was then)
OUT=gpurun_out/r9p; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python scripts/ab_multi.py C3 --sizes 500000,m3/8 --steps 10 -- base base:SAGE_HIP_SHORT_DIVISIONS=1 base base:SAGE_HIP_SHORT_DIVISIONS=1 > $OUT/ab_C3.log 2>&1; grep RESULT -B1 $OUT/ab_C3.log | cut -c1-120
timeout 900 python scripts/ab_multi.py C3T --sizes 500000 --steps 10 -- base base:SAGE_HIP_SHORT_DIVISIONS=1 > $OUT/ab_C3T.log 2>&1; grep RESULT -B1 $OUT/ab_C3T.log | cut -c1-120
grep -h "md5" $OUT/ab_C3.log $OUT/ab_C3T.log | sed 's/.*md5/md5/' | sort | uniq -c
