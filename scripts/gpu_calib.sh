#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known-byte patterns (scripts/calib_traffic.hip); run through gpurun from the repo root.
# usage: scripts/gpu_calib.sh [tag]
TAG=${1:-r02_calib}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
BIN=scripts/calib_traffic
[ -x $BIN ] || hipcc --offload-arch=gfx950 -O3 -o $BIN scripts/calib_traffic.hip
timeout 120 $BIN > $OUT/calib_plain.log 2>&1; echo "plain rc=$?"
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum"; do
  d=$OUT/pmc_$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $pass -d $d -o p -- $BIN > $d.log 2>&1; echo "pmc [$pass] rc=$?"
done
rocprofv3 -L > $OUT/counters_available.txt 2>&1
python scripts/calib_summary.py $OUT/calib_plain.log $OUT/pmc_* > $OUT/calibration.md 2>&1
cat $OUT/calibration.md
find $OUT -name '*.db' -delete
