#!/bin/bash
# round 5, call y: the GPU suite on the final tree (light-end refinement of the mass plan in bench.py / cli.py)
OUT=gpurun_out/r5y; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
