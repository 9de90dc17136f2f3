#!/bin/bash
# r6m: C5's retry pass spends 10 ms in the lane-per-query replay (102 000 queries, 64 per wavefront, as long as its longest stream).
# With the wavefront-per-query replay 2.5x cheaper per offer: every query by wavefront (SAGE_HIP_REPLAY_WAVE_MAX) against the split.
OUT=gpurun_out/r6m; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python scripts/ab_multi.py C5 --sizes 200000 --steps 3 -- base "base:SAGE_HIP_REPLAY_WAVE_MAX=4000000" "base:SAGE_HIP_REPLAY_WAVE_MAX=131072" > $OUT/ab_C5.log 2>&1; echo "ab C5 rc=$?"
grep RESULT -B1 $OUT/ab_C5.log
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 3 -- base "base:SAGE_HIP_REPLAY_WAVE_MAX=4000000" > $OUT/ab_C4.log 2>&1; echo "ab C4 rc=$?"
grep RESULT -B1 $OUT/ab_C4.log
