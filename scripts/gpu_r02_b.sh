#!/bin/bash
OUT=gpurun_out/r02b; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; tail -1 $OUT/build.log
timeout 1500 python -m pytest tests -m gpu -q --durations=12 -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -40 $OUT/pytest.log
( time timeout 900 python bench.py ) > $OUT/bench_C3.json 2> $OUT/bench_C3.err; echo "bench rc=$?"; cat $OUT/bench_C3.json; tail -5 $OUT/bench_C3.err
timeout 600 python bench.py --config C4 --no-traffic > $OUT/bench_C4.json 2> $OUT/bench_C4.err; echo "bench C4 rc=$?"; cat $OUT/bench_C4.json; tail -3 $OUT/bench_C4.err
