#!/bin/bash
# r7y: heaviest precursors first (SAGE_HIP_SCHED_DESC=1) at every shard size of C3, on C2 and C3T; two parts up to 131 072 spectra
OUT=gpurun_out/r7y; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python scripts/ab_multi.py C3 --sizes b1/8,b6/8,b0/4,b0/2,500000 --steps 30 -- base base:SAGE_HIP_SCHED_DESC=1 > $OUT/ab_C3.log 2>&1; grep RESULT -B1 $OUT/ab_C3.log
timeout 1500 python scripts/ab_multi.py C2 --sizes 50000 --steps 40 -- base base:SAGE_HIP_SCHED_DESC=1 > $OUT/ab_C2.log 2>&1; grep RESULT -B1 $OUT/ab_C2.log
timeout 1500 python scripts/ab_multi.py C3T --sizes b3/8,500000 --steps 20 -- base base:SAGE_HIP_SCHED_DESC=1 > $OUT/ab_C3T.log 2>&1; grep RESULT -B1 $OUT/ab_C3T.log
