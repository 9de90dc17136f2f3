#!/bin/bash
# r7f: is prelim_kernel's one-workgroup-per-spectrum launch (500 000 workgroups of one wavefront) a cost?  Capped grids striding over the batch.
OUT=gpurun_out/r7f; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python scripts/ab_multi.py C3 --sizes 500000 --steps 20 -- base base:SAGE_HIP_PRELIM_GRID=5120 base:SAGE_HIP_PRELIM_GRID=10240 base:SAGE_HIP_PRELIM_GRID=40960 base:SAGE_HIP_PRELIM_GRID=163840 > $OUT/ab_C3.log 2>&1; grep RESULT -B1 $OUT/ab_C3.log
timeout 300 python scripts/phase_clocks.py C3 131072 > $OUT/phase_clocks.txt 2>&1; tail -4 $OUT/phase_clocks.txt
