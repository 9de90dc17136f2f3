#!/bin/bash
# SQ counters of rescore_kernel's two instances (IEEE divisions / short divisions) on 131 072 C3 spectra: does the instance that is
# 1-2 % slower really execute fewer vector instructions, and is its vector ALU busy for fewer cycles?
OUT=gpurun_out/r6f; mkdir -p $OUT
CMD="python bench.py --config C3 --spectra 131072 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-extras"
for V in ieee short; do
  if [ $V = short ]; then export SAGE_HIP_SHORT_DIVISIONS=1; else unset SAGE_HIP_SHORT_DIVISIONS; fi
  echo "== $V (a)"; scripts/prof_pmc.sh r6fpmc "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" $CMD 2>&1 | grep -E "^(rescore|pmc)"
  echo "== $V (b)"; scripts/prof_pmc.sh r6fpmc "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" $CMD 2>&1 | grep -E "^(rescore|pmc)"
done | tee $OUT/rescore_instances_pmc.txt
