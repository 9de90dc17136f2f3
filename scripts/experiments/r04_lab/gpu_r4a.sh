#!/bin/bash
# round 4, first GPU call: the GPU suite on the new build (correctly rounded ln, no real call in the narrow kernels), then one A/B of
# library builds on the C3 step at the strong-scaling shard sizes, then a kernel timeline of the 62 500-spectrum step
export TMPDIR=/tmp
OUT=gpurun_out/r4a; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -15 $OUT/pytest_gpu.log
timeout 900 python scripts/ab_multi.py C3 --sizes 62500,125000,500000 --steps 30 -- r3 base rw4 rw6 pw6 base:SAGE_HIP_WAYS=1 r3:SAGE_HIP_WAYS=1 > $OUT/ab.txt 2>&1; echo "ab rc=$?"
cat $OUT/ab.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/tr -o t -- python bench.py --config C3 --spectra 62500 --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --no-extras > $OUT/trace.log 2>&1; echo "trace rc=$?"
python - <<PY
import sqlite3, glob
db = glob.glob("$OUT/tr/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
rows = con.execute("select name, start, end from kernels order by start").fetchall()
def short(n): return n.replace("sagehip::(anonymous namespace)::","").split("(")[0].replace("void ","")[:44]
last = rows[-16:]
t0 = last[0][1]
for n, s, e in last:
    print(f"{short(n):<46} start {(s-t0)/1e3:9.1f} us  end {(e-t0)/1e3:9.1f}  dur {(e-s)/1e3:8.1f} us")
PY
rm -rf $OUT/tr
