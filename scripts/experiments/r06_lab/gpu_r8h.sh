#!/bin/bash
# r8h: kernel traces of C4 with the queue kernel and with the build before (q5): which kernel pays for it?
OUT=gpurun_out/r8h; mkdir -p $OUT; export TMPDIR=/tmp
for L in base q5; do
  LIB=$PWD/sage_amd/libsage_hip.so; [ $L = q5 ] && LIB=$PWD/sage_amd/libsage_hip_q5.so
  SAGE_HIP_LIB=$LIB timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_$L -o t -- python bench.py --config C4 --steps 4 --warmup 2 --no-cpu-baseline --no-traffic --no-extras > $OUT/trace_$L.log 2>&1; echo "trace $L rc=$?"
  python - <<PY
import sqlite3, glob
db = glob.glob("$OUT/trace_$L/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = con.execute(f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3*count(*) desc").fetchall()
for n, c, a in rows[:12]:
    print("$L", n.replace("sagehip::(anonymous namespace)::", "")[:60].ljust(60), c, round(a, 1), "us")
PY
  rm -rf $OUT/trace_$L
done
