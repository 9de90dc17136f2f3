#!/bin/bash
# A/B of kernel build variants on one config: scripts/ab_libs.sh <config> <steps> name1 name2 ...   ("base" = libsage_hip.so)
CFG=${1:-C3}; STEPS=${2:-10}; shift 2
for name in "$@"; do
  lib=$PWD/sage_amd/libsage_hip_$name.so; [ "$name" = base ] && lib=$PWD/sage_amd/libsage_hip.so
  SAGE_HIP_LIB=$lib timeout 200 python bench.py --config $CFG --steps $STEPS --warmup 3 --no-cpu-baseline --no-traffic --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$name', round(d['value']), d['roofline']['kernel_ms'], d['config']['psms_per_step_rank0'])"
done
