"""Ad-hoc: one host-to-host call (sage_hip_score_batch) of a configuration with SAGE_HIP_TIMING=1: the pipeline's host-side
timeline on stderr.  usage: python scripts/h2h_trace.py C3"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SAGE_HIP_TIMING"] = "1"
import bench  # noqa: E402
from sage_amd.api import DeviceDatabase, Scorer  # noqa: E402
from sage_amd.workloads import CONFIGS, build_host_db, scorer_params  # noqa: E402

cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
host = build_host_db(cfg, peptides_only=True)
batch, _ = bench.generate_workload(cfg, host, cfg["spectra"])
dev = DeviceDatabase(host, 0, build_on_device=True)
scorer = Scorer(dev, scorer_params(cfg))
locked = batch.page_locked()
for _ in range(3):
    scorer.score(locked)
print("---- timed call", file=sys.stderr, flush=True)
t0 = time.perf_counter()
scorer.score(locked)
print(f"call: {(time.perf_counter() - t0) * 1e3:.3f} ms", flush=True)
