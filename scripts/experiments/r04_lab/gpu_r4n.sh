#!/bin/bash
# round 4, pass n: the FULL GPU suite on the inline-tie build (+ the epilogue's counter reset), smoke, and the step times
OUT=gpurun_out/r4n; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --durations=8 ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -16 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 600 python scripts/ab_multi.py C3 --sizes 62500,125000,500000 --steps 40 -- base > $OUT/ab_C3.txt 2>&1; cat $OUT/ab_C3.txt
