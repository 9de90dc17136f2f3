#!/bin/bash
# r6v: rescore_kernel's in-line tie replay (C3T: 63 % of the spectra) with the all-lanes root replacement — C3T and C3 against round 5's
# library; tie tests first
OUT=gpurun_out/r6v; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "equal or tie or twin or exact" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest.log
timeout 1500 python scripts/ab_multi.py C3T --sizes 500000 --steps 10 -- base r5 > $OUT/ab_C3T.log 2>&1; grep RESULT -B1 $OUT/ab_C3T.log
timeout 1500 python scripts/ab_multi.py C3 --sizes 500000 --steps 20 -- base r5 > $OUT/ab_C3.log 2>&1; grep RESULT -B1 $OUT/ab_C3.log
