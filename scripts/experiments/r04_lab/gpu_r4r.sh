#!/bin/bash
# round 4, pass r: late arguments from the kernarg segment + the LDS header (base) against the commit before (h1); the full GPU suite
OUT=gpurun_out/r4r; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x ) > $OUT/pytest_parity.log 2>&1; echo "parity rc=$?"; tail -3 $OUT/pytest_parity.log
timeout 600 python scripts/ab_multi.py C3 --sizes 62500,500000 --steps 40 -- h1 base > $OUT/ab_C3.txt 2>&1; cat $OUT/ab_C3.txt
( timeout 1200 python -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
timeout 300 python scripts/phase_clocks.py C3 131072 > $OUT/phase_C3.txt 2>&1; tail -2 $OUT/phase_C3.txt | head -1
