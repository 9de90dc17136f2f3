"""Ad-hoc GPU probe: the host-side timeline of one sage_hip_score_batch call (SAGE_HIP_TIMING=1 prints it on stderr).
usage: python scripts/h2h_timeline.py [config] [n_spectra]"""
import os
import sys
import time

os.environ["SAGE_HIP_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from sage_amd.api import DeviceDatabase, Scorer
from sage_amd.workloads import CONFIGS, build_host_db, scorer_params

cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 500000
host = build_host_db(cfg, peptides_only=True)
batch, _ = bench.generate_workload(cfg, host, min(n, cfg["spectra"]))
scorer = Scorer(DeviceDatabase(host, 0, build_on_device=True), scorer_params(cfg))
locked = batch.page_locked()
for rep in range(3):
    sys.stderr.write(f"==== call {rep}\n")
    t0 = time.perf_counter()
    scorer.score(locked)
    sys.stderr.write(f"==== call {rep}: {(time.perf_counter() - t0) * 1e3:.3f} ms, {batch.n / (time.perf_counter() - t0) / 1e6:.2f} M spectra/s\n")
