#!/bin/bash
# r6d: count kernel — wave totals re-read in the rare overflow loop (vector spills 24 -> 18, no scratch access left in the tile
# loop); against round 5 and against four cells per thread in flight (-DSAGE_TILE8_CELLS=4).  C4 at 20 000 and 100 000 spectra.
OUT=gpurun_out/r6d; mkdir -p $OUT; export TMPDIR=/tmp
for C in C4 C5; do
  timeout 1500 python scripts/ab_multi.py $C --sizes 20000 --steps 4 -- base r5 c4 > $OUT/ab_$C.log 2>&1; echo "ab $C rc=$?"
  grep RESULT -B1 $OUT/ab_$C.log
done
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 3 -- base r5 > $OUT/ab_C4_full.log 2>&1; echo "ab C4 full rc=$?"
grep RESULT -B1 $OUT/ab_C4_full.log
