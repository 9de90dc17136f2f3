#!/bin/bash
# round 5, call q: count kernel with the bisected owner search and one-compare ranges against the build before; tile tests
OUT=gpurun_out/r5q; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_scale.py -m gpu -q -x -k "tile or wide or open or c4 or c5 or chimera or five_thousand or report_psms_beyond" ) > $OUT/pytest_tile.log 2>&1; echo "pytest tile rc=$?"; tail -3 $OUT/pytest_tile.log
timeout 900 python scripts/ab_multi.py C4 --sizes 100000 --steps 6 -- prev base > $OUT/c4.txt 2>&1; cat $OUT/c4.txt
timeout 900 python scripts/ab_multi.py C5 --sizes 200000 --steps 6 -- prev base > $OUT/c5.txt 2>&1; cat $OUT/c5.txt
