#!/bin/bash
OUT=gpurun_out/r02d; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; tail -1 $OUT/build.log
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_scale.py -q -x -k "tile or open or wide or c4 or c5 or chimera or twin or equal" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 $OUT/pytest.log
for C in C4 C5; do
  timeout 300 python bench.py --config $C --no-traffic --no-cpu-baseline --no-extras > $OUT/bench_$C.json 2> $OUT/bench_$C.err; echo "bench $C rc=$?"
  python -c "
import json,sys
d=json.loads(open('$OUT/bench_$C.json').read()); print('$C', round(d['value']), d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['psms_per_step_rank0'], d['roofline']['routing'])" 2>&1 | tail -1
done
timeout 200 python scripts/tile_probe.py 4000 open > $OUT/tile_probe.log 2>&1; tail -4 $OUT/tile_probe.log
