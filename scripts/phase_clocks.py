"""Ad-hoc GPU probe: per-phase shader cycles of the narrow preliminary kernel and the rescoring kernel (their PROF instances).
usage: python scripts/phase_clocks.py [config] [n_spectra]"""
import ctypes as C
import os
import sys

import numpy as np

os.environ["SAGE_HIP_PHASE_CLOCKS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from sage_amd import _lib as L
from sage_amd.api import DeviceDatabase, Scorer
from sage_amd.workloads import CONFIGS, build_host_db, scorer_params

cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
host = build_host_db(cfg, peptides_only=True)
batch, _ = bench.generate_workload(cfg, host, min(n, cfg["spectra"]))
scorer = Scorer(DeviceDatabase(host, 0, build_on_device=True), scorer_params(cfg))
db = scorer.upload(batch)
reps = 3
for _ in range(reps):
    scorer.score_resident(db)
out = np.zeros(32, np.uint64)
L.check(L.load().sage_hip_debug_phase_cycles(scorer._h, L.as_ptr(out, C.c_uint64)))
nn = reps * batch.n
print("prelim  cycles/spectrum: staging %d search %d match %d trim %d output %d" % tuple(out[:5] // nn))
print("prelim  per spectrum: offers %.1f  potential %.1f  queries %.2f" % tuple(out[5:8] / nn))
r = out[8:16] // nn
print("rescore cycles/spectrum: setup(peaks, candidate records) %d | peak table + bitmap %d | filter %d | heavy candidates %d | "
      "lanes' hits %d | ln + hyperscore %d | rank %d | record + store %d   (sum %d)" % (r[0], r[5], r[6], r[7], r[1], r[2], r[3], r[4], r.sum()))
print(scorer.last_timing())
