#!/bin/bash
# post-search rescoring: bench line + rocprofv3 kernel trace (run through gpurun from the repo root)
set -u
N=${1:-1000000}
OUT=gpurun_out/r01_rescore
mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python bench.py --rescore-psms $N --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json; tail -3 $OUT/bench.err
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof/trace -o trace -- python bench.py --rescore-psms $N --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_trace.log 2>&1; echo "trace rc=$?"
python profiles/summarize_rocprof.py r01_rescore RESCORE $(find $OUT/prof/trace -name '*.db' | head -1) > $OUT/summary.txt 2>&1
cp profiles/r01_rescore_rocprof_summary.txt $OUT/ 2>/dev/null
rm -rf $OUT/prof
head -40 $OUT/r01_rescore_rocprof_summary.txt
