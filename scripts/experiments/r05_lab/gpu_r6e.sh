#!/bin/bash
# the other configurations on the final tree, timing only (no CPU baseline, no counter passes): nothing moved with the new instance in the library
OUT=gpurun_out/r6e; mkdir -p $OUT; export TMPDIR=/tmp
for C in C2 C3T C4 C5; do
  timeout 600 python bench.py --config $C --no-cpu-baseline --no-traffic --no-extras > $OUT/bench_$C.json 2> $OUT/bench_$C.err; echo "bench $C rc=$?"
  python - <<PY
import json
j = json.loads([l for l in open("$OUT/bench_$C.json") if l.startswith("{")][-1])
print("$C", round(j["value"]), "spectra/s", round(j["ms_per_step"], 3), "ms/step", j["roofline"].get("kernel_ms"), j.get("parity", {}).get("spectra_checked"))
PY
done
