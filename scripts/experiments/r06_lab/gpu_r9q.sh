#!/bin/bash
# r9q: schedule records (DevBatchView::sched: a block's spectrum, peak range, charge, m/z, isolation window in one 32-byte record in
# schedule order, one trip instead of order[b] + five random reads) in prelim_kernel / rescore_kernel, against SAGE_HIP_NO_SCHED=1
OUT=gpurun_out/r9q; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
A=base; B=base:SAGE_HIP_NO_SCHED=1
timeout 900 python scripts/ab_multi.py C3 --sizes 500000,m3/8 --steps 10 -- $B $A $B $A > $OUT/ab_C3.log 2>&1; grep RESULT -B1 $OUT/ab_C3.log | cut -c1-120
timeout 900 python scripts/ab_multi.py C3T --sizes 500000 --steps 10 -- $B $A > $OUT/ab_C3T.log 2>&1; grep RESULT -B1 $OUT/ab_C3T.log | cut -c1-120
timeout 900 python scripts/ab_multi.py C2 --sizes 50000 --steps 20 -- $B $A $B $A > $OUT/ab_C2.log 2>&1; grep RESULT -B1 $OUT/ab_C2.log | cut -c1-120
timeout 900 python scripts/ab_multi.py C5 --sizes 200000 --steps 4 -- $B $A > $OUT/ab_C5.log 2>&1; grep RESULT -B1 $OUT/ab_C5.log | cut -c1-120
timeout 900 python scripts/ab_multi.py C4 --sizes 100000 --steps 4 -- $B $A > $OUT/ab_C4.log 2>&1; grep RESULT -B1 $OUT/ab_C4.log | cut -c1-120
grep -h "md5" $OUT/ab_*.log | sed 's/.*md5/md5/' | sort | uniq -c
