#!/bin/bash
# round 4, pass ae: the GPU suite against the experiments build (the gated fused / one-launch cases run); cooperative-lookup thresholds once more
OUT=gpurun_out/r4ae; mkdir -p $OUT; export TMPDIR=/tmp
( SAGE_HIP_LIB=$PWD/sage_amd/libsage_hip_exp.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q ) > $OUT/pytest_exp.log 2>&1; echo "pytest(exp) rc=$?"; tail -4 $OUT/pytest_exp.log
timeout 600 python scripts/ab_multi.py C3 --sizes 500000 --steps 40 -- base c10 c14 c10l3 base > $OUT/ab_C3.txt 2>&1; cat $OUT/ab_C3.txt
