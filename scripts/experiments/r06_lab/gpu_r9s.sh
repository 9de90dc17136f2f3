#!/bin/bash
# r9s: the C3 bench line and kernel trace of the final build (early peak requests in rescore_kernel) — the stations of
# scripts/gpu_r6_evidence.sh for C3 only (the round's GPU budget ended here: the other configurations' lines, the SQ counters and
# the phase clocks in profiles/ are of the pass one commit earlier, without the early requests)
TAG=r06; C=C3; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 100 python bench.py --config $C ) > $OUT/${TAG}_${C}_bench.json.new 2> $OUT/bench_$C.err; rc=$?; echo "bench $C rc=$rc"; tail -4 $OUT/bench_$C.err | grep real
if [ $rc -eq 0 ] && grep -q '^{' $OUT/${TAG}_${C}_bench.json.new; then mv $OUT/${TAG}_${C}_bench.json.new $OUT/${TAG}_${C}_bench.json; else exit 1; fi
timeout 60 rocprofv3 --kernel-trace --stats -d $OUT/trace_$C -o t -- python bench.py --config $C --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-extras > $OUT/trace_$C.log 2>&1; echo "trace $C rc=$?"
DB=$(find $OUT/trace_$C -name '*.db' | head -1)
if [ -n "$DB" ]; then python profiles/summarize_rocprof.py ${TAG} $C $DB $OUT/${TAG}_${C}_bench.json > $OUT/${TAG}_${C}_rocprof_summary.txt.new 2>&1 && mv $OUT/${TAG}_${C}_rocprof_summary.txt.new $OUT/${TAG}_${C}_rocprof_summary.txt; fi
rm -rf $OUT/trace_$C
python - <<PY
import json
j = json.loads([l for l in open("$OUT/${TAG}_${C}_bench.json") if l.startswith("{")][-1])
r = j["roofline"]
json.dump({"config": "$C", "workload": j["config"]["workload"], "parity": j["parity"]}, open("$OUT/${TAG}_${C}_full_parity.json", "w"), indent=1)
print("$C", round(j["value"]), "spectra/s", round(j["ms_per_step"], 3), "ms/step", "kernel_ms", r["kernel_ms"], "frac", round(r["frac"], 4), "issue", r.get("frac_issue_slots"), "parity", j["parity"]["spectra_checked"], j["parity"]["psms"], j["parity"]["md5_of_gpu_records"])
PY
