#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r3tl; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/tr -o t -- python bench.py --config C3 --spectra 62500 --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --no-extras > $OUT/trace.log 2>&1; echo "trace rc=$?"
python - <<PY
import sqlite3, glob
db = glob.glob("$OUT/tr/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
# per-dispatch durations of the last step: list kernels in time order for the final 30 dispatches
rows = con.execute("select name, start, end from kernels order by start").fetchall()
names = [r[0].replace("sagehip::(anonymous namespace)::","").split("(")[0].replace("void ","")[:40] for r in rows]
last = rows[-12:]
t0 = last[0][1]
for (n, s, e), nm in zip(last, names[-12:]):
    print(f"{nm:<42} start {(s-t0)/1e3:9.1f} us  dur {(e-s)/1e3:8.1f} us")
PY
rm -rf $OUT/tr
