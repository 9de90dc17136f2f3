#!/bin/bash
# r6b: hit records with lane-private record slots in the preliminary kernel and batched loads of the omitted ions in the
# rescoring kernel — phase clocks (records on / off) and the C3 step A/B.
OUT=gpurun_out/r6b; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python scripts/phase_clocks.py C3 131072 > $OUT/clocks_hits.log 2>&1; tail -n 4 $OUT/clocks_hits.log
SAGE_HIP_NO_HITS=1 timeout 600 python scripts/phase_clocks.py C3 131072 > $OUT/clocks_nohits.log 2>&1; tail -n 4 $OUT/clocks_nohits.log
timeout 900 python scripts/ab_multi.py C3 --sizes 62500,500000 --steps 20 -- base "base:SAGE_HIP_NO_HITS=1" > $OUT/ab.log 2>&1; echo "ab rc=$?"
grep RESULT -B1 $OUT/ab.log
