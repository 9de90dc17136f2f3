#!/bin/bash
# rescore kernel iteration: parity suites, then C3 / C5 / C2 bench lines and the rescore phase probe
OUT=gpurun_out/r02j; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_scale.py tests/test_gpu_rescore.py -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -n 3 $OUT/pytest.log
for C in C3 C5 C2; do
  timeout 300 python bench.py --config $C --no-traffic --no-cpu-baseline --no-extras > $OUT/bench_$C.json 2> $OUT/bench_$C.err; echo "bench $C rc=$?"
  python -c "
import json,sys
d=json.loads(open('$OUT/bench_$C.json').read()); print('$C', round(d['value']), d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['psms_per_step_rank0'], d['roofline']['routing'])" 2>&1 | tail -n 1
done
timeout 400 python scripts/phase_probe.py C3 65536 2>&1 | grep -v amdgpu.ids | tail -n 2
