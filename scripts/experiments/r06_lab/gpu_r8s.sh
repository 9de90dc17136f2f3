#!/bin/bash
# r8s: kernel trace of C5 and C4 on the current build (per launch: the first pass and the retry pass apart)
OUT=gpurun_out/r8s; mkdir -p $OUT; export TMPDIR=/tmp
for C in C5 C4; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_$C -o t -- python bench.py --config $C --steps 3 --warmup 2 --no-cpu-baseline --no-traffic --no-extras > $OUT/trace_$C.log 2>&1; echo "trace $C rc=$?"
  python - <<PY
import sqlite3, glob
db = glob.glob("$OUT/trace_$C/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = con.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
# the last step: from the last prelim_kernel launch on
idx = max(i for i, r in enumerate(rows) if 'prelim_kernel' in r[0])
t0 = rows[idx][1]
for n, a, b in rows[idx:]:
    print("$C", n.replace("_ZN7sagehip12_GLOBAL__N_1", "")[:44].ljust(44), "start %8.1f us  dur %8.1f us" % ((a - t0) / 1e3, (b - a) / 1e3))
PY
  rm -rf $OUT/trace_$C
done
