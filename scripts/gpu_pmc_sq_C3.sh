#!/bin/bash
# SQ counters of the narrow-search kernels on C3 (131072 spectra; averages are per dispatch, two dispatches per step: the full
# pass and the small retry pass, so per-spectrum figures are ~2x the averages / 131072)
export TMPDIR=/tmp
OUT=gpurun_out/r03pmc; mkdir -p $OUT
CMD="python bench.py --config C3 --spectra 131072 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-extras"
scripts/prof_pmc.sh r03pmc "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" $CMD 2>&1 | grep -E "^(rescore|prelim_k|pmc)" | tee $OUT/pmc_C3_a.txt
scripts/prof_pmc.sh r03pmc "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" $CMD 2>&1 | grep -E "^(rescore|prelim_k|pmc)" | tee $OUT/pmc_C3_b.txt
