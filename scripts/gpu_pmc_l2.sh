#!/bin/bash
# L2 / fabric counters of the narrow-search kernels on C3 (131072 spectra), with and without the XCD-aware schedule remap.
# usage: gpurun -- scripts/gpu_pmc_l2.sh [tag]
export TMPDIR=/tmp
TAG=${1:-r3pmc}; OUT=gpurun_out/$TAG; mkdir -p $OUT
CMD="python bench.py --config C3 --spectra 131072 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-extras"
for CH in 0 1024; do
  export SAGE_HIP_XCD_CHUNK=$CH
  echo "== SAGE_HIP_XCD_CHUNK=$CH" | tee -a $OUT/pmc_l2.txt
  scripts/prof_pmc.sh $TAG "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" $CMD 2>&1 | grep -E "^(rescore|prelim_k|narrow|pmc)" | tee -a $OUT/pmc_l2.txt
  scripts/prof_pmc.sh $TAG "FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" $CMD 2>&1 | grep -E "^(rescore|prelim_k|narrow|pmc)" | tee -a $OUT/pmc_l2.txt
done
