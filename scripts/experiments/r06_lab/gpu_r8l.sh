#!/bin/bash
# r8l: the tie statistic in 16 striped counters: parity + config-scale + bench-contract tests, C3T / C3 against the build before (q6)
OUT=gpurun_out/r8l; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest.log
timeout 1500 python scripts/ab_multi.py C3T --sizes 500000 --steps 10 -- q6 base > $OUT/ab_C3T.log 2>&1; grep RESULT -B1 $OUT/ab_C3T.log
timeout 1500 python scripts/ab_multi.py C3 --sizes 500000 --steps 10 -- q6 base > $OUT/ab_C3.log 2>&1; grep RESULT -B1 $OUT/ab_C3.log
