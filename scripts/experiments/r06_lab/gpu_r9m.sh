#!/bin/bash
# r9m: the dense work list as the default build (phase B fetches the ions): the whole GPU suite incl. the new route test, then against the walk (-DSAGE_DENSE_HITS=0)
OUT=gpurun_out/r9m; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
timeout 900 python scripts/ab_multi.py C3 --sizes 500000,m3/8 --steps 10 -- walk base walk base > $OUT/ab_C3.log 2>&1; grep RESULT -B1 $OUT/ab_C3.log | cut -c1-120
timeout 900 python scripts/ab_multi.py C2 --sizes 50000 --steps 20 -- walk base > $OUT/ab_C2.log 2>&1; grep RESULT -B1 $OUT/ab_C2.log | cut -c1-120
timeout 900 python scripts/ab_multi.py C5 --sizes 200000 --steps 4 -- walk base > $OUT/ab_C5.log 2>&1; grep RESULT -B1 $OUT/ab_C5.log | cut -c1-120
timeout 900 python scripts/ab_multi.py C4 --sizes 100000 --steps 4 -- walk base > $OUT/ab_C4.log 2>&1; grep RESULT -B1 $OUT/ab_C4.log | cut -c1-120
