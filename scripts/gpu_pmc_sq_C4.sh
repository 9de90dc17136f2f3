#!/bin/bash
# SQ counters of the large-window kernels on C4 (16384 spectra) -> profiles/r02_C4_pmc_sq_{a,b}.txt
export TMPDIR=/tmp
OUT=gpurun_out/pmc_C4; mkdir -p $OUT
CMD="python bench.py --config C4 --spectra 16384 --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-extras"
PMC_TIMEOUT=120 scripts/prof_pmc.sh pmc_C4 "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" $CMD 2>&1 | grep -E "^(tile_|pmc)" | tee $OUT/pmc_C4_a.txt
PMC_TIMEOUT=120 scripts/prof_pmc.sh pmc_C4 "SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" $CMD 2>&1 | grep -E "^(tile_|pmc)" | tee $OUT/pmc_C4_b.txt
