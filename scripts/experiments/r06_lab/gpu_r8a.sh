#!/bin/bash
# r8a: heaviest-first for narrow batches + two parts up to 196 608 spectra as defaults: the whole GPU suite, the C3 shards, C4 / C5 unchanged
OUT=gpurun_out/r8a; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest.log
timeout 1500 python scripts/ab_multi.py C3 --sizes b0/8,b3/8,b7/8,b0/4,b0/2,500000 --steps 30 -- base > $OUT/ab_C3.log 2>&1; grep RESULT -B1 $OUT/ab_C3.log
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 4 -- base > $OUT/ab_C4.log 2>&1; grep RESULT -B1 $OUT/ab_C4.log
