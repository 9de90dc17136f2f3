#!/bin/bash
# instruction-cache / wait counters: search_kernel (one launch, two roles) against prelim_kernel + rescore_kernel, C3, 131072 spectra
export TMPDIR=/tmp
TAG=${1:-r3s}; OUT=gpurun_out/$TAG; mkdir -p $OUT
CMD="python bench.py --config C3 --spectra 131072 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-extras"
for TL in 1 0; do
  export SAGE_HIP_TWO_LAUNCHES=$TL SAGE_HIP_SEARCH_LAG=100000000
  echo "== SAGE_HIP_TWO_LAUNCHES=$TL" | tee -a $OUT/pmc.txt
  PMC_TIMEOUT=100 scripts/prof_pmc.sh $TAG "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_IFETCH" $CMD 2>&1 | grep -E "^(rescore|prelim_k|narrow|search|pmc)" | tee -a $OUT/pmc.txt
  PMC_TIMEOUT=100 scripts/prof_pmc.sh $TAG "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM" $CMD 2>&1 | grep -E "^(rescore|prelim_k|narrow|search|pmc)" | tee -a $OUT/pmc.txt
done
