#!/bin/bash
# bench lines for every BASELINE.json configuration (no profiling)
set -u
OUT=gpurun_out/${1:-benchall}; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
for CFG in ${2:-C3 C4 C5 C2}; do
  timeout 900 python bench.py --config $CFG > $OUT/bench_$CFG.json 2> $OUT/bench_$CFG.err; echo "$CFG rc=$?"
  cat $OUT/bench_$CFG.json; tail -3 $OUT/bench_$CFG.err
done
cp profiles/algorithmic_bytes.json $OUT/ 2>/dev/null
