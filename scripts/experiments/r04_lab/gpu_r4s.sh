#!/bin/bash
# round 4, pass s: prelim_kernel's arguments phase by phase from the kernarg segment (base) against the commit before (h2)
OUT=gpurun_out/r4s; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x ) > $OUT/pytest_parity.log 2>&1; echo "parity rc=$?"; tail -3 $OUT/pytest_parity.log
timeout 600 python scripts/ab_multi.py C3 --sizes 62500,500000 --steps 40 -- h2 base base:SAGE_HIP_WAYS=2 > $OUT/ab_C3.txt 2>&1; cat $OUT/ab_C3.txt
timeout 600 python scripts/ab_multi.py C2 --sizes 500000 --steps 20 -- h2 base > $OUT/ab_C2.txt 2>&1; cat $OUT/ab_C2.txt
( timeout 1200 python -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
