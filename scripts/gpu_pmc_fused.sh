#!/bin/bash
# SQ / instruction-cache counters of the fused narrow kernel against the separate kernels, C3, 131072 spectra.
export TMPDIR=/tmp
TAG=${1:-r3f}; OUT=gpurun_out/$TAG; mkdir -p $OUT
rocprofv3 --list-avail 2>/dev/null | grep -o -E "\b(SQC?_[A-Z_0-9]*(ICACHE|IFETCH|INST_LEVEL|BUSY)[A-Z_0-9]*)\b" | sort -u | head -40 > $OUT/avail_icache.txt
cat $OUT/avail_icache.txt | tr '\n' ' '; echo
CMD="python bench.py --config C3 --spectra 131072 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-extras"
for NF in 1 0; do
  export SAGE_HIP_NO_FUSED=$NF
  echo "== SAGE_HIP_NO_FUSED=$NF" | tee -a $OUT/pmc.txt
  PMC_TIMEOUT=100 scripts/prof_pmc.sh $TAG "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" $CMD 2>&1 | grep -E "^(rescore|prelim_k|narrow|pmc)" | tee -a $OUT/pmc.txt
  PMC_TIMEOUT=100 scripts/prof_pmc.sh $TAG "SQ_IFETCH SQ_WAIT_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" $CMD 2>&1 | grep -E "^(rescore|prelim_k|narrow|pmc)" | tee -a $OUT/pmc.txt
done
