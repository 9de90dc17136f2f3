#!/bin/bash
# call z: the >64-queries-per-spectrum case of test_large_window_tile_kernel, then the parity file
mkdir -p gpurun_out/r6z
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "large_window" > gpurun_out/r6z/pytest_large_window.txt 2>&1
tail -5 gpurun_out/r6z/pytest_large_window.txt
