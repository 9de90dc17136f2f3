#!/bin/bash
# round 5, call h: 32 against 16 mass blocks per rank; the count kernel's window query through pep_lut on C4 / C5
OUT=gpurun_out/r5h; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python scripts/ab_multi.py C3 --sizes c0/8,c1/8,c2/8,c3/8,c4/8,c5/8,c6/8,c7/8 --steps 40 -- base:AB_TIMING_EVERY=4,AB_BLOCKS=32 base:AB_TIMING_EVERY=4,AB_BLOCKS=16 > $OUT/c3_shards.txt 2>&1; cat $OUT/c3_shards.txt
timeout 900 python scripts/ab_multi.py C5 --sizes 200000 --steps 6 -- base > $OUT/c5.txt 2>&1; cat $OUT/c5.txt
timeout 900 python scripts/ab_multi.py C4 --sizes 100000 --steps 6 -- base > $OUT/c4.txt 2>&1; cat $OUT/c4.txt
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_scale.py -m gpu -q -x -k "tile or wide or open or c4 or c5 or chimera" ) > $OUT/pytest_tile.log 2>&1; echo "pytest tile rc=$?"; tail -3 $OUT/pytest_tile.log
