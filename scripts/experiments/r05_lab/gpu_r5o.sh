#!/bin/bash
# round 5, call o: what GRBM_GUI_ACTIVE / SQ_BUSY_CYCLES rows look like per dispatch (the issue-slot fraction's denominator)
OUT=gpurun_out/r5o; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES -d /tmp/pmcx -o p -- python $GRAFT_REPO_ROOT/bench.py --config C3 --spectra 65536 --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-extras > $GRAFT_REPO_ROOT/$OUT/cmd.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import sqlite3, glob
db = glob.glob("/tmp/pmcx/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
print([r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")][:60])
cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
print(cols)
for row in con.execute("select * from counters_collection where kernel_name like '%rescore_kernel%' limit 24"):
    print(row)
print("-- grouped")
for row in con.execute("select kernel_name, counter_name, count(*), min(value), max(value), avg(value), sum(value) from counters_collection where kernel_name like '%rescore_kernel%' or kernel_name like '%prelim_kernel%' group by kernel_name, counter_name"):
    print(row)
PY
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/ktx -o t -- python bench.py --config C3 --spectra 65536 --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-extras > $OUT/cmd2.log 2>&1
python - <<PY
import sqlite3, glob
db = glob.glob("/tmp/ktx/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
for name, s, e in con.execute("select name, start, end from kernels where name like '%rescore_kernel%' order by start"):
    print(name[:60], (e - s) / 1e3, "us")
PY
