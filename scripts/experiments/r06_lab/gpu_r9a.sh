#!/bin/bash
# r9a: the u8 count kernel with 2 cells per thread in flight (7 vector spills instead of 18; units above 1 024 cells take a synchronous second round)
OUT=gpurun_out/r9a; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 4 -- base tc2 > $OUT/ab_C4.log 2>&1; grep RESULT -B1 $OUT/ab_C4.log
timeout 1500 python scripts/ab_multi.py C5 --sizes 200000 --steps 4 -- base tc2 > $OUT/ab_C5.log 2>&1; grep RESULT -B1 $OUT/ab_C5.log
