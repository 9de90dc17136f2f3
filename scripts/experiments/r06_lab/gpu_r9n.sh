#!/bin/bash
# r9n: the heavy candidates' lookups through the dense work list as well (-DSAGE_DENSE_HEAVY=1: lib dh) against the default build
OUT=gpurun_out/r9n; mkdir -p $OUT; export TMPDIR=/tmp
SAGE_HIP_LIB=$PWD/sage_amd/libsage_hip_dh.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x --no-header -p no:cacheprovider > $OUT/pytest_dh.log 2>&1; tail -1 $OUT/pytest_dh.log
timeout 600 python scripts/ab_multi.py C3 --sizes 500000,m3/8 --steps 10 -- base dh base dh > $OUT/ab_C3.log 2>&1; grep RESULT -B1 $OUT/ab_C3.log | cut -c1-120
timeout 600 python scripts/ab_multi.py C3T --sizes 500000 --steps 10 -- base dh > $OUT/ab_C3T.log 2>&1; grep RESULT -B1 $OUT/ab_C3T.log | cut -c1-120
timeout 600 python scripts/ab_multi.py C2 --sizes 50000 --steps 20 -- base dh > $OUT/ab_C2.log 2>&1; grep RESULT -B1 $OUT/ab_C2.log | cut -c1-120
CMD="python bench.py --config C3 --spectra 131072 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-extras"
echo "== lib_dh"; SAGE_HIP_LIB=$PWD/sage_amd/libsage_hip_dh.so scripts/prof_pmc.sh r9n "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES" $CMD 2>&1 | grep -E "^(rescore|pmc)"
