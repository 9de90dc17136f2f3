#!/bin/bash
# r6p: count kernel — scan / emit / clear shared over the window's slot range of the tile (narrow windows: balanced wavefronts).
# Full GPU parity file, the config-scale suite; C5 and C4 at full size; C5 phase clocks.
OUT=gpurun_out/r6p; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_scale.py -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -n 4 $OUT/pytest.log
timeout 1500 python scripts/ab_multi.py C5 --sizes 200000 --steps 3 -- base r5 > $OUT/ab_C5.log 2>&1; grep RESULT -B1 $OUT/ab_C5.log
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 3 -- base > $OUT/ab_C4.log 2>&1; grep RESULT -B1 $OUT/ab_C4.log
timeout 900 python scripts/tile_phase_cfg.py C5 20000 2>&1 | tail -11
