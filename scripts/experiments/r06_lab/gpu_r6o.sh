#!/bin/bash
# r6o: count kernel — every precursor-window query of a spectrum searched up front, a wavefront each.  Tile tests; C5 and C4 at full size.
OUT=gpurun_out/r6o; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "tile or open or wide or chimera or large or asymmetric or exact or equal or isotope or unknown" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -n 4 $OUT/pytest.log
timeout 1500 python scripts/ab_multi.py C5 --sizes 200000 --steps 3 -- base > $OUT/ab_C5.log 2>&1; grep RESULT -B1 $OUT/ab_C5.log
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 3 -- base > $OUT/ab_C4.log 2>&1; grep RESULT -B1 $OUT/ab_C4.log
timeout 900 python scripts/tile_phase_cfg.py C5 20000 2>&1 | tail -11
