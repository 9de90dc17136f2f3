#!/bin/bash
# round 5, call w: the driver's 8-rank launch line at FULL size on one GPU (gloo rehearsal backend: the ranks share the device) — the
# mass exchange, the timed region, the ordered gather and the single-GPU identity check over all 500 000 C3 spectra.  The spectra/s
# of this line mean nothing (eight processes time-slice one GPU); what counts is that it runs and that `identical_to_single_gpu` is true.
OUT=gpurun_out/r5w; mkdir -p $OUT; export TMPDIR=/tmp
( time SAGE_BENCH_BACKEND=gloo timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 8 --steps 5 --warmup 2 --no-extras ) > $OUT/rehearsal8.json 2> $OUT/rehearsal8.err; echo "rc=$?"; tail -3 $OUT/rehearsal8.err
python - <<PY
import json
ls = [l for l in open("$OUT/rehearsal8.json") if l.startswith("{")]
j = json.loads(ls[-1])
print(j["n_gpus"], j["scaling"], j["config"]["spectra_total"], j["config"]["parallelism"], j["sharding"])
PY
