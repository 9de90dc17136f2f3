#!/bin/bash
# r7a: prelim_kernel's probe — a window's bounds worked out once per batch and kept in LDS (wb), + one division for a symmetric ppm
# tolerance (wbs), + that division in the short form behind a wave-wide range check (base), against the build before (head).
OUT=gpurun_out/r7a; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest.log
timeout 1200 python scripts/ab_multi.py C3 --sizes 62500,500000 --steps 20 -- head wb wbs base > $OUT/ab_C3.log 2>&1; grep RESULT -B1 $OUT/ab_C3.log
timeout 1200 python scripts/ab_multi.py C2 --sizes 50000 --steps 30 -- head base > $OUT/ab_C2.log 2>&1; grep RESULT -B1 $OUT/ab_C2.log
