"""Phase clocks of the large-window count kernel on a BASELINE configuration (GPU probe, not a pytest).
usage: python scripts/tile_phase_cfg.py [C4|C5] [n_spectra]"""
import ctypes as C
import os
import sys
import time

import numpy as np

os.environ["SAGE_HIP_PHASE_CLOCKS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sage_amd import _lib as L  # noqa: E402
from sage_amd.api import DeviceDatabase, Scorer  # noqa: E402
from sage_amd.workloads import CONFIGS, build_host_db, scorer_params  # noqa: E402

cfg_name = sys.argv[1] if len(sys.argv) > 1 else "C4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
cfg = CONFIGS[cfg_name]
params = scorer_params(cfg)
host = build_host_db(cfg, peptides_only=True)
batch, _ = bench.generate_workload(cfg, host, n)
dev = DeviceDatabase(host, 0, build_on_device=True)
scorer = Scorer(dev, params)
db = scorer.upload(batch)
scorer.score_resident(db)
t0 = time.perf_counter()
f, c = scorer.score_resident(db)
dt = time.perf_counter() - t0
t = scorer.last_timing()
out = np.zeros(32, np.uint64)
L.check(L.load().sage_hip_debug_phase_cycles(scorer._h, L.as_ptr(out, C.c_uint64)))
nn = 2.0 * batch.n  # (two calls)
names = ["query+setup", "wait cells + apply (stream)", "barrier 1", "publish + cell loads", "threshold + scan + emit", "clear", "barrier 2", "(wait for cells only)"]
print(cfg_name, "spectra", batch.n, "ms/step", round(dt * 1e3, 2), {k: (round(v, 2) if isinstance(v, float) else v) for k, v in t.items()})
tot = float(out[16:23].sum())
for i in range(7):
    print(f"  {names[i]:<30} {out[16 + i] / nn:>10.0f} cycles per spectrum  {100.0 * out[16 + i] / tot:5.1f} %")
print(f"  {names[7]:<30} {out[23] / nn:>10.0f} cycles per spectrum (part of the stream row)")
print("  byte counters per spectrum: table words", out[29] / nn, "cells", out[30] / nn, "candidate words", out[31] / nn)
