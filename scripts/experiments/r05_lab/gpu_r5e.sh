#!/bin/bash
# round 5, call e: PSM counts stored by the kernels into the caller's array in multi-part steps; parts / schedule knobs on the shards
OUT=gpurun_out/r5e; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q -x ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
timeout 1200 python scripts/ab_multi.py C3 --sizes c0/8,c7/8,c3/4,c1/2,500000 --steps 40 -- base base:SAGE_HIP_WAYS=3 base:SAGE_HIP_WAYS=4 base:SAGE_HIP_SCHED_DESC=1 base:SAGE_HIP_XCD_CHUNK=256 base:SAGE_HIP_WAYS=1 > $OUT/c3_knobs.txt 2>&1; cat $OUT/c3_knobs.txt
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/tr -o t -- python bench.py --config C3 --slice 0/8 --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --no-extras > $OUT/trace.log 2>&1; echo "trace rc=$?"
python - <<PY
import sqlite3, glob
db = glob.glob("$OUT/tr/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
rows = con.execute("select name, start, end from kernels order by start").fetchall()
names = [r[0].replace("sagehip::(anonymous namespace)::","").split("(")[0].replace("void ","")[:40] for r in rows]
last = rows[-16:]
t0 = last[0][1]
for (n, s, e), nm in zip(last, names[-16:]):
    print(f"{nm:<42} start {(s-t0)/1e3:9.1f} us  end {(e-t0)/1e3:9.1f} us  dur {(e-s)/1e3:8.1f} us")
PY
rm -rf $OUT/tr
