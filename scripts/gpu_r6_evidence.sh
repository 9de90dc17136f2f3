#!/bin/bash
# Round 6 evidence pass on one GPU box: full GPU suite, smoke, then per config the bench line (whole-workload parity, live PMC
# traffic + issue-slot counters, one-sample CPU thread table) and a rocprofv3 kernel trace summarised into profiles/; the SQ
# counters and phase clocks of C3; the count kernel's phase clocks on C4 / C5.  The shard-size table is scripts/experiments/r06_lab/gpu_r6f.sh's.
# usage: gpurun -- scripts/gpu_r6_evidence.sh [configs...]
TAG=r06; CFGS=${@:-C3 C2 C4 C5 C3T}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --durations=6 ) > $OUT/${TAG}_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -12 $OUT/${TAG}_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
for C in $CFGS; do
  ( time timeout 900 python bench.py --config $C ) > $OUT/${TAG}_${C}_bench.json 2> $OUT/bench_$C.err; echo "bench $C rc=$?"; tail -4 $OUT/bench_$C.err | grep real
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_$C -o t -- python bench.py --config $C --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-extras > $OUT/trace_$C.log 2>&1; echo "trace $C rc=$?"
  python profiles/summarize_rocprof.py ${TAG} $C $(find $OUT/trace_$C -name '*.db' | head -1) $OUT/${TAG}_${C}_bench.json > $OUT/${TAG}_${C}_rocprof_summary.txt 2>&1
  rm -rf $OUT/trace_$C
  python - <<PY
import json
j = json.loads([l for l in open("$OUT/${TAG}_${C}_bench.json") if l.startswith("{")][-1])
r = j["roofline"]
json.dump({"config": "$C", "workload": j["config"]["workload"], "parity": j["parity"]}, open("$OUT/${TAG}_${C}_full_parity.json", "w"), indent=1)
print("$C", round(j["value"]), "spectra/s", round(j["ms_per_step"], 3), "ms/step", "kernel_ms", r["kernel_ms"], "frac", round(r["frac"], 4), "issue", r.get("frac_issue_slots"),
      "traffic", r.get("traffic"), "routing", r.get("routing"), "parity", j["parity"]["spectra_checked"], j["parity"]["psms"], j["parity"]["md5_of_gpu_records"])
PY
done
CMD="python bench.py --config C3 --spectra 131072 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-extras"
scripts/prof_pmc.sh ${TAG}pmc "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" $CMD 2>&1 | grep -E "^(rescore|prelim_k|narrow|pmc)" | tee $OUT/${TAG}_C3_pmc_sq_a.txt
scripts/prof_pmc.sh ${TAG}pmc "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" $CMD 2>&1 | grep -E "^(rescore|prelim_k|narrow|pmc)" | tee $OUT/${TAG}_C3_pmc_sq_b.txt
timeout 300 python scripts/phase_clocks.py C3 131072 > $OUT/${TAG}_C3_phase_clocks.txt 2>&1; tail -4 $OUT/${TAG}_C3_phase_clocks.txt
timeout 300 python scripts/tile_phase_cfg.py C4 20000 > $OUT/${TAG}_C4_count_phase_clocks.txt 2>&1; tail -11 $OUT/${TAG}_C4_count_phase_clocks.txt
timeout 300 python scripts/tile_phase_cfg.py C5 40000 > $OUT/${TAG}_C5_count_phase_clocks.txt 2>&1; tail -11 $OUT/${TAG}_C5_count_phase_clocks.txt
