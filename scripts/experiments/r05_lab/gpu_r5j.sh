#!/bin/bash
# round 5, call j: peptides beyond 1023 residues through the general instances; phase clocks of the count kernel on C4 / C5
OUT=gpurun_out/r5j; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "beyond_1023 or report_psms_beyond" ) > $OUT/pytest_long.log 2>&1; echo "pytest long rc=$?"; tail -25 $OUT/pytest_long.log
timeout 600 python scripts/tile_phase_cfg.py C4 20000 > $OUT/phase_c4.txt 2>&1; tail -12 $OUT/phase_c4.txt
timeout 600 python scripts/tile_phase_cfg.py C5 40000 > $OUT/phase_c5.txt 2>&1; tail -12 $OUT/phase_c5.txt
