#!/bin/bash
# usage: scripts/gpu_trace.sh <tag> <config> [extra bench args]: rocprofv3 kernel trace of a short bench run -> per-kernel table
TAG=$1; CFG=$2; shift; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_$CFG -o t -- python bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-extras "$@" > $OUT/trace_$CFG.log 2>&1; echo "trace rc=$?"
python - <<PY
import sqlite3, glob
db = glob.glob("$OUT/trace_$CFG/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
print("kernel calls total_us avg_us pct")
for name, calls, total, avg, pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    short = name.replace("sagehip::(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    if pct > 0.3: print(f"{short:<60} {calls:>6} {total/1e3:>12.1f} {avg/1e3:>10.2f} {pct:>6.2f}")
PY
rm -rf $OUT/trace_$CFG
