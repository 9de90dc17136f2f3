#!/bin/bash
# the GPU suite, smoke() and the default bench line on the final tree (the short-division instance in the library, switched off)
OUT=gpurun_out/r6c; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --durations=6 ) > $OUT/r05_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -12 $OUT/r05_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
( time timeout 900 python bench.py ) > $OUT/bench_C3.json 2> $OUT/bench_C3.err; echo "bench rc=$?"; tail -3 $OUT/bench_C3.err; cut -c1-600 $OUT/bench_C3.json
