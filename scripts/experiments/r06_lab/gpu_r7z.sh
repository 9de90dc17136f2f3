#!/bin/bash
# r7z: heaviest-first as the default: narrow parity tests; the number of parts at 4 GPUs' shard size again; C4 / C5 / C3 unchanged?
OUT=gpurun_out/r7z; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_scale.py -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest.log
timeout 1500 python scripts/ab_multi.py C3 --sizes b0/4,b1/4,b0/3 --steps 30 -- base base:SAGE_HIP_WAYS=2 > $OUT/ab_b04.log 2>&1; grep RESULT -B1 $OUT/ab_b04.log
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 4 -- base:SAGE_HIP_SCHED_DESC=0 base > $OUT/ab_C4.log 2>&1; grep RESULT -B1 $OUT/ab_C4.log
timeout 1500 python scripts/ab_multi.py C5 --sizes 200000 --steps 4 -- base:SAGE_HIP_SCHED_DESC=0 base > $OUT/ab_C5.log 2>&1; grep RESULT -B1 $OUT/ab_C5.log
