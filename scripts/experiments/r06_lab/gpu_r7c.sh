#!/bin/bash
# r7c: prelim_kernel, now short of its issue ceiling (0.61): 6 wavefronts per SIMD; 192 / 256 windows per batch (one batch for most
# spectra), with and without the next batch's table reads in flight under this batch's cells
OUT=gpurun_out/r7c; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest.log
timeout 1500 python scripts/ab_multi.py C3 --sizes 62500,500000 --steps 20 -- base w6 p3 p3np p4np > $OUT/ab_C3.log 2>&1; grep RESULT -B1 $OUT/ab_C3.log
