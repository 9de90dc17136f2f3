#!/bin/bash
# second sheet of the instruction calibration: the integer / compare / conversion / LDS instructions rescore_kernel is made of
mkdir -p gpurun_out/r5z
hipcc --offload-arch=gfx950 -O3 -o /tmp/calib_valu scripts/calib_valu.hip || exit 1
CALIB_SECOND_SHEET=1 timeout 300 /tmp/calib_valu > gpurun_out/r5z/calib_second_sheet.txt 2>&1
cat gpurun_out/r5z/calib_second_sheet.txt
