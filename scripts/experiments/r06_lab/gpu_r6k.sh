#!/bin/bash
# r6k: the heap replay of a large-window query with the root replacement done by all lanes at once (two ds_bpermute + two ballots
# per entering offer instead of a six-level readlane / writelane loop) — tile / exact parity tests, C4 and C5 at full size against
# round 5 (the retry column is the replay's).
OUT=gpurun_out/r6k; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "tile or open or wide or chimera or large or asymmetric or exact or equal" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -n 5 $OUT/pytest.log
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 3 -- base r5 > $OUT/ab_C4.log 2>&1; echo "ab C4 rc=$?"
grep RESULT -B1 $OUT/ab_C4.log
timeout 1500 python scripts/ab_multi.py C5 --sizes 200000 --steps 3 -- base r5 > $OUT/ab_C5.log 2>&1; echo "ab C5 rc=$?"
grep RESULT -B1 $OUT/ab_C5.log
