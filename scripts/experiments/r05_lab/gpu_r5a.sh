#!/bin/bash
# round 5, first GPU call: the GPU suite on the new build (asymmetric tolerances, WCAP 8192 one-PSM ties, mass-sharded CLI), the
# scalar-pipe calibration, the C3 step on mass-contiguous vs input-contiguous shards of an 8-GPU run, and the bench line with
# whole-workload parity.
OUT=gpurun_out/r5a; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q -x --durations=8 ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_gpu.log
timeout 200 scripts/calib_valu > $OUT/valu_calibration.md 2>&1; echo "calib rc=$?"; cat $OUT/valu_calibration.md
timeout 900 python scripts/ab_multi.py C3 --sizes 62500,i0/8,i3/8,i7/8,m0/8,m1/8,m3/8,m5/8,m7/8,500000 --steps 40 -- base > $OUT/shards.txt 2>&1; cat $OUT/shards.txt
( time timeout 900 python bench.py --config C3 ) > $OUT/r05_C3_bench.json 2> $OUT/bench_C3.err; echo "bench rc=$?"; tail -5 $OUT/bench_C3.err
python - <<PY
import json
j = json.loads([l for l in open("$OUT/r05_C3_bench.json") if l.startswith("{")][-1])
print(round(j["value"]), j["ms_per_step"], j["parity"], j["cpu_baseline"]["threads_table"], j["roofline"]["kernel_ms"])
PY
