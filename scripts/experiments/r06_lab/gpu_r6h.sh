#!/bin/bash
# r6h: the small tiles' position table in succinct form (occupancy + rank per 32 cells, run starts of the non-empty cells) —
# narrow parity tests, then the C3 and C2 step against round 5's row-major table.
OUT=gpurun_out/r6h; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -n 5 $OUT/pytest.log
timeout 900 python scripts/ab_multi.py C3 --sizes 62500,500000 --steps 20 -- base r5 > $OUT/ab_C3.log 2>&1; echo "ab rc=$?"
grep RESULT -B1 $OUT/ab_C3.log
timeout 900 python scripts/ab_multi.py C2 --sizes 50000 --steps 30 -- base r5 > $OUT/ab_C2.log 2>&1; echo "ab rc=$?"
grep RESULT -B1 $OUT/ab_C2.log
