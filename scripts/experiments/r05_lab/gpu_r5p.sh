#!/bin/bash
# round 5, call p: the C3 line again with the counter-based issue-slot fraction; the contract tests
OUT=gpurun_out/r05; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_bench_contract.py -m gpu -q -x ) > $OUT/pytest_contract.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_contract.log
for C in C3 C4; do
( time timeout 900 python bench.py --config $C ) > $OUT/r05_${C}_bench.json 2> $OUT/bench_$C.err; echo "bench $C rc=$?"
python - <<PY
import json
j = json.loads([l for l in open("$OUT/r05_${C}_bench.json") if l.startswith("{")][-1])
r = j["roofline"]
json.dump({"config": "$C", "workload": j["config"]["workload"], "parity": j["parity"]}, open("$OUT/r05_${C}_full_parity.json", "w"), indent=1)
print("$C", round(j["value"]), j["ms_per_step"], r["kernel_ms"], "issue", r["issue"], "parity", j["parity"]["spectra_checked"], j["parity"]["md5_of_gpu_records"])
PY
done
