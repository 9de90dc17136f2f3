#!/bin/bash
# r7b: SQ counters of prelim_kernel with the window bounds in LDS (131 072 C3 spectra)
OUT=gpurun_out/r7b; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python bench.py --config C3 --spectra 131072 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-extras"
scripts/prof_pmc.sh r7bpmc "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" $CMD 2>&1 | grep -E "^(rescore|prelim_k|pmc)" | tee $OUT/pmc_sq_a.txt
scripts/prof_pmc.sh r7bpmc "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" $CMD 2>&1 | grep -E "^(rescore|prelim_k|pmc)" | tee $OUT/pmc_sq_b.txt
timeout 300 python scripts/phase_clocks.py C3 131072 > $OUT/phase_clocks.txt 2>&1; tail -4 $OUT/phase_clocks.txt
