#!/bin/bash
# r9o: wave priority by phase (s_setprio; -DSAGE_SETPRIO_R / _P = 1: staging at priority 3, 2: the tail at priority 2, 3: both) against the
# hardware's oldest-first arbitration.  No arithmetic changes: the records' md5 must be equal across the variants.
OUT=gpurun_out/r9o; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python scripts/ab_multi.py C3 --sizes 500000,m3/8 --steps 10 -- base r1 r2 r3 p1 p2 p3 base > $OUT/ab_C3.log 2>&1; grep RESULT -B1 $OUT/ab_C3.log | cut -c1-120
grep -i "md5" $OUT/ab_C3.log | sort | uniq -c | cut -c1-120
