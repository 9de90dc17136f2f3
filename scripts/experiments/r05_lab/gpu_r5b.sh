#!/bin/bash
# round 5, call b: per-mass cost profile of C3 (32 equal-count mass slices), the torchrun rehearsals, the timeline of a
# mass-slice step (bench --slice 3/8)
OUT=gpurun_out/r5b; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python scripts/shard_cost_probe.py C3 32 30 > $OUT/shard_cost.txt 2>&1; grep -E "^(SLICE|FEATURES)" $OUT/shard_cost.txt
( timeout 600 python -m pytest tests/test_bench_contract.py tests/test_cli_io.py -m gpu -q -x ) > $OUT/pytest_b.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_b.log
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/tr -o t -- python bench.py --config C3 --slice 3/8 --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --no-extras > $OUT/trace.log 2>&1; echo "trace rc=$?"; tail -2 $OUT/trace.log | cut -c1-400
python - <<PY
import sqlite3, glob
db = glob.glob("$OUT/tr/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
rows = con.execute("select name, start, end from kernels order by start").fetchall()
names = [r[0].replace("sagehip::(anonymous namespace)::","").split("(")[0].replace("void ","")[:40] for r in rows]
last = rows[-27:]
t0 = last[0][1]
for (n, s, e), nm in zip(last, names[-27:]):
    print(f"{nm:<42} start {(s-t0)/1e3:9.1f} us  end {(e-t0)/1e3:9.1f} us  dur {(e-s)/1e3:8.1f} us")
PY
rm -rf $OUT/tr
( timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_bench_contract.py --deselect tests/test_cli_io.py ) > $OUT/pytest_rest.log 2>&1; echo "pytest rest rc=$?"; tail -8 $OUT/pytest_rest.log
