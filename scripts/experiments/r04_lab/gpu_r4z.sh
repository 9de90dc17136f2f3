#!/bin/bash
# round 4, pass z: device-side timeline (kernels + copies) of one host-to-host call
OUT=gpurun_out/r4z; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/tr -o t -- python scripts/h2h_timeline.py C3 500000 > $OUT/run.log 2>&1; echo "rc=$?"
python - <<PY
import sqlite3, glob
db = glob.glob("$OUT/tr/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if 'copy' in t.lower() or 'kernel' in t.lower()][:20])
ev = []
for n, s, e in con.execute("select name, start, end from kernels"):
    ev.append((s, e, n.replace("sagehip::(anonymous namespace)::","").split("(")[0].replace("void ","")[:40]))
try:
    cols = [r[1] for r in con.execute("pragma table_info(memory_copies)")]
    print(cols)
    for r in con.execute("select name, start, end, size from memory_copies"):
        ev.append((r[1], r[2], f"COPY {r[0]} {r[3]/1e6:.2f} MB"))
except Exception as ex:
    print("copies:", ex)
ev.sort()
t_end = ev[-1][1]
last = [x for x in ev if x[0] > t_end - 12_000_000]
t0 = last[0][0]
for s, e, n in last:
    print(f"{(s-t0)/1e3:9.1f} {(e-t0)/1e3:9.1f} {(e-s)/1e3:8.1f} us  {n}")
PY
rm -rf $OUT/tr
