#!/bin/bash
# round 4, pass x: host to host — the schedule sort beside the peak copies, the window estimate beside the first chunk's staging (base) vs the commit before (h5)
OUT=gpurun_out/r4x; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python scripts/ab_multi.py C3 --sizes 500000 --steps 30 --h2h -- h5 base h5 base > $OUT/ab_C3.txt 2>&1; cat $OUT/ab_C3.txt
( timeout 900 python -m pytest tests -m gpu -q -x ) > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
