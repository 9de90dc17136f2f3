#!/bin/bash
# round 4 evidence for C3: the bench line (live PMC traffic, CPU thread table), a kernel trace, the SQ counters of the narrow
# kernels, the step at the strong-scaling shard sizes, the VALU issue-rate calibration
TAG=r04; OUT=gpurun_out/r04c3; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 900 python bench.py --config C3 ) > $OUT/${TAG}_C3_bench.json 2> $OUT/bench_C3.err; echo "bench C3 rc=$?"; tail -4 $OUT/bench_C3.err | grep real
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_C3 -o t -- python bench.py --config C3 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-extras > $OUT/trace_C3.log 2>&1; echo "trace rc=$?"
python profiles/summarize_rocprof.py ${TAG} C3 $(find $OUT/trace_C3 -name '*.db' | head -1) $OUT/${TAG}_C3_bench.json > $OUT/${TAG}_C3_rocprof_summary.txt 2>&1
rm -rf $OUT/trace_C3
CMD="python bench.py --config C3 --spectra 131072 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-extras"
scripts/prof_pmc.sh r04c3 "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" $CMD 2>&1 | grep -E "^(rescore|prelim_k|tie|pmc)" | tee $OUT/${TAG}_C3_pmc_sq_a.txt
scripts/prof_pmc.sh r04c3 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" $CMD 2>&1 | grep -E "^(rescore|prelim_k|tie|pmc)" | tee $OUT/${TAG}_C3_pmc_sq_b.txt
scripts/gpu_shard_sizes.sh > $OUT/${TAG}_shard_sizes.txt 2>&1; cat $OUT/${TAG}_shard_sizes.txt
scripts/gpu_calib_valu.sh r04c3 > /dev/null 2>&1; cat $OUT/valu_calibration.md
head -c 1500 $OUT/${TAG}_C3_bench.json; echo
head -30 $OUT/${TAG}_C3_rocprof_summary.txt
