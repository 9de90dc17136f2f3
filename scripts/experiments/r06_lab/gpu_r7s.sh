#!/bin/bash
# r7s: rank locate, third version (marks' set and offset not live across the tile loop; arrays at constant distances), tests + C4 / C5 against q4
OUT=gpurun_out/r7s; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_scale.py -q -x -k "large_window or tile or open or c4 or c5 or wide or chimera or edge" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest.log
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 5 -- q4 base > $OUT/ab_C4.log 2>&1; grep RESULT -B1 $OUT/ab_C4.log
timeout 1500 python scripts/ab_multi.py C5 --sizes 200000 --steps 5 -- q4 base > $OUT/ab_C5.log 2>&1; grep RESULT -B1 $OUT/ab_C5.log
timeout 300 python scripts/tile_phase_cfg.py C4 20000 > $OUT/C4_count_phase_clocks.txt 2>&1; tail -11 $OUT/C4_count_phase_clocks.txt
