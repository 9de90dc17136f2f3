#!/bin/bash
# usage: scripts/prof_cmd.sh <tag> <command...> : rocprofv3 kernel trace of a command, prints the per-kernel stats
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- "$@" > $OUT/cmd.log 2>&1
tail -4 $OUT/cmd.log
python - <<PY
import sqlite3, glob
db = glob.glob("$OUT/trace/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
for name, calls, total, avg, pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    short = name.replace("sagehip::(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    print(f"{short:<45} calls={calls:<4} total_us={total:>12.1f} avg_us={avg:>10.2f} pct={pct:>6.2f}")
PY
rm -rf $OUT/trace
