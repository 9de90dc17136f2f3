#!/bin/bash
# r9e: the cooperative matching's thresholds once more on the round's kernel: 1 heavy candidate at most; from 8 / 10 / 14 hits per chunk (12 is the default)
OUT=gpurun_out/r9e; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python scripts/ab_multi.py C3 --sizes 500000 --steps 10 -- base cm1 mh8 mh10 mh14 > $OUT/ab_C3.log 2>&1; grep RESULT -B1 $OUT/ab_C3.log
timeout 1500 python scripts/ab_multi.py C3T --sizes 500000 --steps 10 -- base cm1 mh8 mh10 mh14 > $OUT/ab_C3T.log 2>&1; grep RESULT -B1 $OUT/ab_C3T.log
