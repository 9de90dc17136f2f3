#!/bin/bash
# r9r: rescore_kernel requests a spectrum's first 192 peaks in front of the branches on its status (with schedule records the peak
# range is known one trip earlier; the compiler does not hoist loads above the early returns) against -DSAGE_EARLY_PEAKS=0 (lib np);
# then the whole GPU suite on the default build
OUT=gpurun_out/r9r; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python scripts/ab_multi.py C3 --sizes 500000,m3/8 --steps 10 -- np base np base > $OUT/ab_C3.log 2>&1; grep RESULT -B1 $OUT/ab_C3.log | cut -c1-120
grep -h "md5" $OUT/ab_C3.log | sed 's/.*md5/md5/' | sort | uniq -c
timeout 400 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
