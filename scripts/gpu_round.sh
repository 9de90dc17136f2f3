#!/bin/bash
# One GPU-box pass: [parity tests,] bench line, rocprofv3 kernel trace + separate PMC passes for one config.
# usage: scripts/gpu_round.sh <tag> <config> [notest]      (run through gpurun from the repo root)
set -u
TAG=${1:-r01}; CFG=${2:-C3}; NOTEST=${3:-}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
if [ -z "$NOTEST" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
  tail -5 $OUT/pytest_gpu.log
fi
timeout 900 python bench.py --config $CFG > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json; tail -3 $OUT/bench.err
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof/trace -o trace -- python bench.py --config $CFG --steps 10 --warmup 2 --no-cpu-baseline > $OUT/prof_trace.log 2>&1; echo "trace rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/prof/pmc_fetch -o pmc -- python bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_fetch.log 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/prof/pmc_write -o pmc -- python bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_write.log 2>&1; echo "write rc=$?"
python profiles/summarize_rocprof.py $TAG $CFG $(find $OUT/prof/trace -name '*.db' | head -1) $(find $OUT/prof/pmc_fetch -name '*.db' | head -1) $(find $OUT/prof/pmc_write -name '*.db' | head -1) > $OUT/summary.txt 2>&1
cp profiles/${TAG}_rocprof_summary.txt profiles/traffic.json profiles/algorithmic_bytes.json $OUT/ 2>/dev/null
rm -rf $OUT/prof   # the sqlite traces stay on the box; the text summary is what gets committed
tail -40 $OUT/summary.txt
