#!/bin/bash
# round 5, call s: the count kernel's scan without a prefix sum for wavefronts that have no candidate bit in a tile
OUT=gpurun_out/r5s; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python scripts/ab_multi.py C4 --sizes 100000 --steps 6 -- base skip > $OUT/c4.txt 2>&1; cat $OUT/c4.txt
timeout 900 python scripts/ab_multi.py C5 --sizes 200000 --steps 6 -- base skip > $OUT/c5.txt 2>&1; cat $OUT/c5.txt
( SAGE_HIP_LIB=$PWD/sage_amd/libsage_hip_skip.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tile or wide or open or chimera" ) > $OUT/pytest_tile.log 2>&1; echo "pytest tile rc=$?"; tail -3 $OUT/pytest_tile.log
