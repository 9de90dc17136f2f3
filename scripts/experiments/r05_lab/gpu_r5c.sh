#!/bin/bash
# round 5, call c: GPU suite on the restructured resident step (no fork / join events, per-part epilogue); stream vs probe matching
# on C2 and on the heavy end of C3; the strided-block mass shards; timeline of a shard step
OUT=gpurun_out/r5c; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q -x --durations=5 ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest_gpu.log
timeout 600 python scripts/ab_multi.py C2 --sizes 50000 --steps 40 -- base:SAGE_HIP_NARROW=stream base:SAGE_HIP_NARROW=probe > $OUT/c2_variants.txt 2>&1; cat $OUT/c2_variants.txt
timeout 900 python scripts/ab_multi.py C3 --sizes 62500,b0/8,b3/8,b7/8,m7/8,e31/32,e0/32,500000 --steps 40 -- base base:SAGE_HIP_NARROW=probe > $OUT/c3_shards.txt 2>&1; cat $OUT/c3_shards.txt
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/tr -o t -- python bench.py --config C3 --slice 3/8 --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --no-extras > $OUT/trace.log 2>&1; echo "trace rc=$?"
python - <<PY
import sqlite3, glob
db = glob.glob("$OUT/tr/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
rows = con.execute("select name, start, end from kernels order by start").fetchall()
names = [r[0].replace("sagehip::(anonymous namespace)::","").split("(")[0].replace("void ","")[:40] for r in rows]
last = rows[-21:]
t0 = last[0][1]
for (n, s, e), nm in zip(last, names[-21:]):
    print(f"{nm:<42} start {(s-t0)/1e3:9.1f} us  end {(e-t0)/1e3:9.1f} us  dur {(e-s)/1e3:8.1f} us")
PY
rm -rf $OUT/tr
