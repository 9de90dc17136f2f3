#!/bin/bash
OUT=gpurun_out/r02k; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_scale.py -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -n 3 $OUT/pytest.log
scripts/ab_libs.sh C3 10 base w8 w5
scripts/ab_libs.sh C5 5 base
timeout 300 python scripts/phase_probe.py C3 65536 2>&1 | grep -v amdgpu.ids | tail -n 2
CMD="python bench.py --config C3 --spectra 131072 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-extras"
scripts/prof_pmc.sh r02k "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" $CMD 2>&1 | grep -E "^(rescore|prelim_k|pmc)"
