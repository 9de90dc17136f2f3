#!/bin/bash
# The C3 step on the shard sizes of a strong-scaling run (500k / N), and two scorer handles on one GPU beside the sequential
# figure (bench.py's `concurrent` extra).  usage: gpurun -- scripts/gpu_shard_sizes.sh
export TMPDIR=/tmp
for n in 62500 125000 250000 500000; do
  timeout 300 python bench.py --config C3 --spectra $n --steps 40 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print($n, 'spectra/s', round(d['value']), 'ms/step', round(d['ms_per_step'],3), d['roofline']['kernel_ms'] if d['roofline'] else '', 'concurrent', round(d['concurrent']['value']))"
done
