#!/bin/bash
# r6u: prelim_kernel's owner search by marks + a DPP prefix maximum (round 4's patch, "no change" when the kernel followed its line
# traffic) on the issue-bound kernel of round 6, against the 7-step binary search (-DSAGE_PROBE_OWNER_SCAN=0).  Narrow parity tests first.
OUT=gpurun_out/r6u; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest.log
timeout 1200 python scripts/ab_multi.py C3 --sizes 62500,500000 --steps 20 -- base nos > $OUT/ab_C3.log 2>&1; grep RESULT -B1 $OUT/ab_C3.log
timeout 1200 python scripts/ab_multi.py C2 --sizes 50000 --steps 30 -- base nos > $OUT/ab_C2.log 2>&1; grep RESULT -B1 $OUT/ab_C2.log
