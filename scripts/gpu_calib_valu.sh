#!/bin/bash
# VALU issue-rate calibration (scripts/calib_valu.hip) -> gpurun_out/<tag>/valu_calibration.md
TAG=${1:-r05_calib_valu}; OUT=gpurun_out/$TAG; mkdir -p $OUT
BIN=scripts/calib_valu
[ -x $BIN ] || hipcc --offload-arch=gfx950 -O3 -o $BIN scripts/calib_valu.hip
timeout 120 $BIN > $OUT/valu_calibration.md 2>&1; echo "calib rc=$?"
cat $OUT/valu_calibration.md
