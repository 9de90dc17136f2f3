#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4c; mkdir -p $OUT
timeout 900 python scripts/ab_multi.py C3 --sizes 62500,500000 --steps 30 -- p1 base notie base:SAGE_HIP_NO_FAST_TIES=1 > $OUT/ab_C3.txt 2>&1; echo "ab rc=$?"
cat $OUT/ab_C3.txt
timeout 900 python scripts/ab_multi.py C3T --sizes 62500,500000 --steps 10 -- base > $OUT/ab_C3T.txt 2>&1; echo "ab rc=$?"
cat $OUT/ab_C3T.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "equal_hyperscores or platform_libm or narrow_search" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/tr -o t -- python bench.py --config C3 --spectra 62500 --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --no-extras > $OUT/trace.log 2>&1; echo "trace rc=$?"
python - <<PY
import sqlite3, glob
db = glob.glob("$OUT/tr/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
rows = con.execute("select name, start, end from kernels order by start").fetchall()
def short(n): return n.replace("sagehip::(anonymous namespace)::","").split("(")[0].replace("void ","")[:44]
last = rows[-20:]
t0 = last[0][1]
for n, s, e in last:
    print(f"{short(n):<46} start {(s-t0)/1e3:9.1f} us  end {(e-t0)/1e3:9.1f}  dur {(e-s)/1e3:8.1f} us")
PY
rm -rf $OUT/tr
