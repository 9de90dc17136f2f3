#!/bin/bash
# r7q: SQ counters of the count kernel with the rank locate (C5, 100 000 spectra: compare gpu_r7h.sh)
OUT=gpurun_out/r7q; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python bench.py --config C5 --spectra 100000 --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-extras"
scripts/prof_pmc.sh r7qpmc "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" $CMD 2>&1 | grep -E "^(tile_count8|pmc)" | tee $OUT/pmc_sq_a.txt
scripts/prof_pmc.sh r7qpmc "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" $CMD 2>&1 | grep -E "^(tile_count8|pmc)" | tee $OUT/pmc_sq_b.txt
