#!/bin/bash
# r7e: query_window's one-trip form (both partition points and the edge rule's masses from two reads in flight together: four
# dependent trips to pep_mono become one) against the build before (wbsf); C3 and the open-search configurations (the count kernel
# searches its windows with the same function)
OUT=gpurun_out/r7e; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest.log
timeout 1500 python scripts/ab_multi.py C3 --sizes 62500,500000 --steps 20 -- wbsf base > $OUT/ab_C3.log 2>&1; grep RESULT -B1 $OUT/ab_C3.log
timeout 1500 python scripts/ab_multi.py C5 --sizes 200000 --steps 5 -- wbsf base > $OUT/ab_C5.log 2>&1; grep RESULT -B1 $OUT/ab_C5.log
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 5 -- wbsf base > $OUT/ab_C4.log 2>&1; grep RESULT -B1 $OUT/ab_C4.log
