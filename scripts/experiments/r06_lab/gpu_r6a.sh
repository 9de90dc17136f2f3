#!/bin/bash
# r6a: hit records (prelim RECORD + rescore HITS) — first GPU contact.  Parity suite with the records on (default), then the C3 step
# with records against SAGE_HIP_NO_HITS=1 (md5 of the PSM records must agree), 500 000 and 62 500 spectra.
OUT=gpurun_out/r6a; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -n 15 $OUT/pytest.log
timeout 900 python scripts/ab_multi.py C3 --sizes 62500,500000 --steps 20 -- base "base:SAGE_HIP_NO_HITS=1" > $OUT/ab.log 2>&1; echo "ab rc=$?"
tail -n 30 $OUT/ab.log
