#!/bin/bash
OUT=gpurun_out/r02m; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_scale.py -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -n 3 $OUT/pytest.log
scripts/ab_libs.sh C3 10 base
scripts/ab_libs.sh C2 20 base
scripts/ab_libs.sh C4 5 base
scripts/ab_libs.sh C5 5 base
CMD="python bench.py --config C4 --spectra 16384 --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-extras"
PMC_TIMEOUT=120 scripts/prof_pmc.sh r02m "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" $CMD 2>&1 | grep -E "^(tile_|pmc)" | tee $OUT/pmc_C4_a.txt
PMC_TIMEOUT=120 scripts/prof_pmc.sh r02m "SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" $CMD 2>&1 | grep -E "^(tile_|pmc)" | tee $OUT/pmc_C4_b.txt
