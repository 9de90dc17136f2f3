#!/bin/bash
# r6g: the driver's launch line at 8 and 2 ranks, FULL size, on one GPU (gloo rehearsal backend: the ranks share the device) with
# the file exchange of round 6 — what the mass exchange and the ordered gather cost in seconds (sharding.exchange_by_mass_s /
# gather_s), n_ranks_seen, identity with the single-GPU pass.  The spectra/s of these lines mean nothing.
OUT=gpurun_out/r6g; mkdir -p $OUT; export TMPDIR=/tmp
for N in 8 2; do
  ( time SAGE_BENCH_BACKEND=gloo timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2951$N bench.py --gpus $N --steps 5 --warmup 2 --no-extras ) > $OUT/rehearsal$N.out 2> $OUT/rehearsal$N.err; echo "N=$N rc=$?"; tail -4 $OUT/rehearsal$N.err
  python - <<PY
import json
ls = [l for l in open("$OUT/rehearsal$N.out") if l.startswith("{")]
j = json.loads(ls[-1])
json.dump(j, open("$OUT/rehearsal_${N}ranks.json", "w"), indent=1)
print(j["n_gpus"], j["scaling"], j["config"]["spectra_total"], j["config"]["parallelism"], j["n_ranks_seen"], j["sharding"])
PY
done
