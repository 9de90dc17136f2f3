#!/bin/bash
# One GPU-box pass for the round's evidence: the full GPU test suite, then per config the bench line (live PMC traffic, CPU
# thread table) and a rocprofv3 kernel trace, summarised into profiles/.
# usage: scripts/gpu_round.sh <tag> [configs...]      (run through gpurun from the repo root)
TAG=${1:-r03}; shift; CFGS=${@:-C3 C2 C4 C5}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; tail -1 $OUT/build.log
( time timeout 1500 python -m pytest tests -m gpu -q --durations=10 ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -18 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
for C in $CFGS; do
  ( time timeout 900 python bench.py --config $C ) > $OUT/${TAG}_${C}_bench.json 2> $OUT/bench_$C.err; echo "bench $C rc=$?"; tail -4 $OUT/bench_$C.err | grep real
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_$C -o t -- python bench.py --config $C --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-extras > $OUT/trace_$C.log 2>&1; echo "trace $C rc=$?"
  python profiles/summarize_rocprof.py ${TAG} $C $(find $OUT/trace_$C -name '*.db' | head -1) $OUT/${TAG}_${C}_bench.json > $OUT/${TAG}_${C}_rocprof_summary.txt 2>&1
  rm -rf $OUT/trace_$C
done
