#!/bin/bash
# r9d: how many heavy candidates of a spectrum the wavefront takes together (SAGE_COOP_MAX_LANES 2 / 3 / 4 / 8) on the tie-rich C3T and on C3
OUT=gpurun_out/r9d; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python scripts/ab_multi.py C3T --sizes 500000 --steps 10 -- base cm3 cm4 cm8 > $OUT/ab_C3T.log 2>&1; grep RESULT -B1 $OUT/ab_C3T.log
timeout 1500 python scripts/ab_multi.py C3 --sizes 500000 --steps 10 -- base cm3 cm4 cm8 > $OUT/ab_C3.log 2>&1; grep RESULT -B1 $OUT/ab_C3.log
