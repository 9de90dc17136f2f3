#!/bin/bash
# round 5, call v: PMC traffic of a rank's shard of an 8-GPU run, cut by mass blocks against cut by input position (same spectrum counts)
OUT=gpurun_out/r5v; mkdir -p $OUT; export TMPDIR=/tmp
for BY in mass input; do
  timeout 900 python bench.py --config C3 --slice 3/8 --shard-by $BY --steps 40 --warmup 5 --no-cpu-baseline --no-extras --traffic-timeout 400 > $OUT/slice_$BY.json 2> $OUT/slice_$BY.err; echo "rc=$?"
  python - <<PY
import json
j = json.loads([l for l in open("$OUT/slice_$BY.json") if l.startswith("{")][-1])
r = j["roofline"]
print("$BY", j["config"]["slice"], "ms/step", round(j["ms_per_step"], 4), "kernel_ms", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r["kernel_ms"].items()},
      "traffic bytes per spectrum", r["traffic_bytes_per_spectrum"], r["traffic_source"][:60], "issue", r["issue"] and {k: r["issue"][k] for k in ("valu_per_spectrum", "frac_issue_slots")})
PY
done
