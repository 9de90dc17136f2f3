#!/bin/bash
# round 5, call d: probe as the default matching variant + the exact window check at upload + count-balanced strided blocks;
# the quad layout of the open-search position table (variant build) on C4 / C5 and the tile tests
OUT=gpurun_out/r5d; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q -x --durations=5 ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log
( SAGE_HIP_LIB=$PWD/sage_amd/libsage_hip_quad.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_scale.py -m gpu -q -x -k "tile or wide or open or c4 or c5 or chimera" ) > $OUT/pytest_quad.log 2>&1; echo "pytest quad rc=$?"; tail -4 $OUT/pytest_quad.log
timeout 900 python scripts/ab_multi.py C3 --sizes 62500,c0/8,c1/8,c2/8,c3/8,c4/8,c5/8,c6/8,c7/8,500000 --steps 40 -- base > $OUT/c3_shards.txt 2>&1; cat $OUT/c3_shards.txt
timeout 600 python scripts/ab_multi.py C2 --sizes 50000 --steps 40 -- base > $OUT/c2.txt 2>&1; cat $OUT/c2.txt
timeout 900 python scripts/ab_multi.py C4 --sizes 100000 --steps 6 -- base quad > $OUT/c4_quad.txt 2>&1; cat $OUT/c4_quad.txt
timeout 900 python scripts/ab_multi.py C5 --sizes 200000 --steps 6 -- base quad > $OUT/c5_quad.txt 2>&1; cat $OUT/c5_quad.txt
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/tr -o t -- python bench.py --config C3 --slice 0/8 --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --no-extras > $OUT/trace.log 2>&1; echo "trace rc=$?"
python - <<PY
import sqlite3, glob
db = glob.glob("$OUT/tr/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
rows = con.execute("select name, start, end from kernels order by start").fetchall()
names = [r[0].replace("sagehip::(anonymous namespace)::","").split("(")[0].replace("void ","")[:40] for r in rows]
last = rows[-24:]
t0 = last[0][1]
for (n, s, e), nm in zip(last, names[-24:]):
    print(f"{nm:<42} start {(s-t0)/1e3:9.1f} us  end {(e-t0)/1e3:9.1f} us  dur {(e-s)/1e3:8.1f} us")
PY
rm -rf $OUT/tr
