#!/bin/bash
# r8z: the large tiles' size on the round's count kernel (VERDICT r05 task 3b: 2^16 peptides per tile at two workgroups per compute unit; also 2^14)
OUT=gpurun_out/r8z; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 4 -- base base:SAGE_HIP_TILE_SHIFT=16 base:SAGE_HIP_TILE_SHIFT=14 > $OUT/ab_C4.log 2>&1; grep RESULT -B1 $OUT/ab_C4.log
timeout 1500 python scripts/ab_multi.py C5 --sizes 200000 --steps 4 -- base base:SAGE_HIP_TILE_SHIFT=16 base:SAGE_HIP_TILE_SHIFT=14 > $OUT/ab_C5.log 2>&1; grep RESULT -B1 $OUT/ab_C5.log
