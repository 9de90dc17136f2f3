#!/bin/bash
# round 4, pass t: the peptide-mass position table (precursor-window search in two scalar + two wave-wide reads)
OUT=gpurun_out/r4t; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python scripts/ab_multi.py C3 --sizes 62500,500000 --steps 40 -- base:SAGE_HIP_NO_PEP_LUT=1 base > $OUT/ab_C3.txt 2>&1; cat $OUT/ab_C3.txt
( timeout 1200 python -m pytest tests -m gpu -q -x ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
