#!/bin/bash
# r6c: count kernel with its arguments read from the kernarg segment phase by phase (143 -> 65 scalar spills, 33 -> 24 vector spills)
# and the per-tile suffix sums through DPP — tile parity tests, then C4 / C5 against the round-5 build (libsage_hip_r5.so).
OUT=gpurun_out/r6c; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "tile or open or wide or chimera or large or asymmetric or five_thousand" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -n 5 $OUT/pytest.log
for C in C4 C5; do
  timeout 900 python scripts/ab_multi.py $C --sizes 20000 --steps 4 -- base r5 > $OUT/ab_$C.log 2>&1; echo "ab $C rc=$?"
  grep RESULT -B1 $OUT/ab_$C.log
done
