#!/bin/bash
# r9h: the cooperative matching's threshold on top of the dense work list (8 / 12 / 16 / 24 hits per chunk)
OUT=gpurun_out/r9h; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python scripts/ab_multi.py C3 --sizes 500000 --steps 10 -- base dense dmh8 dmh16 dmh24 dense > $OUT/ab_C3.log 2>&1; grep RESULT -B1 $OUT/ab_C3.log
timeout 900 python scripts/ab_multi.py C3T --sizes 500000 --steps 10 -- dense dmh8 dmh16 dmh24 > $OUT/ab_C3T.log 2>&1; grep RESULT -B1 $OUT/ab_C3T.log
