#!/bin/bash
OUT=gpurun_out/r02h; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_scale.py -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 $OUT/pytest.log
timeout 400 python scripts/phase_probe.py C4 8192 2>&1 | grep -v amdgpu.ids | tail -3
scripts/gpu_trace.sh r02h C4
scripts/gpu_trace.sh r02h C5
