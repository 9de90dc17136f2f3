#!/bin/bash
# parity tests + bench lines for the listed configs (no profiling)
set -u
OUT=gpurun_out/${1:-tb}; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -25 $OUT/pytest_gpu.log
for CFG in ${2:-C3 C4 C5 C2}; do
  timeout 900 python bench.py --config $CFG > $OUT/bench_$CFG.json 2> $OUT/bench_$CFG.err; echo "$CFG rc=$?"
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$CFG.json"))
    r=d["roofline"]; c=d.get("cpu_baseline") or {}
    print("$CFG value %.4g spectra/s  ms/step %.3f  prelim %.3f rescore %.3f  frac %.4f  cpu %.4g  speedup %.1f  pcie %.4g" % (d["value"], d["ms_per_step"], r["kernel_ms"]["prelim"], r["kernel_ms"]["rescore"], r["frac"], c.get("value",0), d.get("speedup_vs_cpu_baseline",0), d["pcie_inclusive_value"]))
    print("   routing", r["routing"], c.get("parity"))
except Exception as e:
    print("ERR", e)
PY
  tail -3 $OUT/bench_$CFG.err
done
cp profiles/algorithmic_bytes.json $OUT/ 2>/dev/null
