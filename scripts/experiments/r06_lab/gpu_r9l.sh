#!/bin/bash
# r9l: dense work list with phase B fetching the items' ions (one word per item from the lanes; dg), + phase C over the matched items (dgm),
# against the first forms (dense: lanes gather their ions in phase A; dense2: + matched-only phase C) and the lane-by-lane walk
OUT=gpurun_out/r9l; mkdir -p $OUT; export TMPDIR=/tmp
for v in dg dgm; do SAGE_HIP_LIB=$PWD/sage_amd/libsage_hip_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x --no-header -p no:cacheprovider > $OUT/pytest_$v.log 2>&1; tail -1 $OUT/pytest_$v.log; done
timeout 900 python scripts/ab_multi.py C3 --sizes 500000 --steps 10 -- base dense dense2 dg dgm base dense dense2 dg dgm > $OUT/ab_C3.log 2>&1; grep RESULT -B1 $OUT/ab_C3.log | cut -c1-120
timeout 900 python scripts/ab_multi.py C3T --sizes 500000 --steps 10 -- base dense dense2 dg dgm > $OUT/ab_C3T.log 2>&1; grep RESULT -B1 $OUT/ab_C3T.log | cut -c1-120
CMD="python bench.py --config C3 --spectra 131072 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-extras"
for v in _dg _dgm; do echo "== lib$v"; SAGE_HIP_LIB=$PWD/sage_amd/libsage_hip$v.so scripts/prof_pmc.sh r9l "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES" $CMD 2>&1 | grep -E "^(rescore|pmc)"; done
