#!/bin/bash
# round 4, pass p: A/B of the cheaper bitmap build + the shared ln pass (base) against the commit before (h0); parity subset; phase clocks
OUT=gpurun_out/r4p; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python scripts/ab_multi.py C3 --sizes 62500,500000 --steps 40 -- h0 base > $OUT/ab_C3.txt 2>&1; cat $OUT/ab_C3.txt
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_scale.py -m gpu -q -x ) > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 300 python scripts/phase_clocks.py C3 131072 > $OUT/phase_C3.txt 2>&1; tail -4 $OUT/phase_C3.txt
