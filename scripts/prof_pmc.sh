#!/bin/bash
# usage: scripts/prof_pmc.sh <tag> "<counters>" <command...> : rocprofv3 --pmc pass, per-kernel average of each counter
TAG=$1; CTRS=$2; shift; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 5 ${PMC_TIMEOUT:-150} rocprofv3 --pmc $CTRS -d $OUT/pmc -o p -- "$@" > $OUT/cmd.log 2>&1 || { echo "pmc pass failed or timed out"; tail -n 3 $OUT/cmd.log; exit 1; }
python - <<PY
import sqlite3, glob
db = glob.glob("$OUT/pmc/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
rows = {}
for name, ctr, n, avg in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
    short = name.replace("sagehip::(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    rows.setdefault(short, {})[ctr] = avg
for k, v in rows.items():
    if k.startswith("__amd"): continue
    print(k, " ".join(f"{c}={x:.4g}" for c, x in sorted(v.items())))
PY
rm -rf $OUT/pmc
