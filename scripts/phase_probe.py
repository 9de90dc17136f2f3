"""Ad-hoc GPU probe (not a pytest): in-kernel phase cycles of the large-window count kernel on a BASELINE config.
usage: python scripts/phase_probe.py [config] [n_spectra]"""
import os, sys, time, numpy as np, ctypes as C
os.environ["SAGE_HIP_PHASE_CLOCKS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sage_amd import _lib as L
from sage_amd.api import DeviceDatabase, Scorer
from sage_amd.workloads import CONFIGS, build_host_db, scorer_params, workload_batch
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
cfg = CONFIGS[name]
host = build_host_db(cfg, peptides_only=True)
batch, _ = workload_batch(cfg, host, 0, n)
dev = DeviceDatabase(host, 0, build_on_device=True)
scorer = Scorer(dev, scorer_params(cfg)); db = scorer.upload(batch)
scorer.score_resident(db)
t0 = time.perf_counter(); f, c = scorer.score_resident(db); dt = time.perf_counter() - t0
t = scorer.last_timing()
print(name, t, "spectra/s %.4g" % (batch.n / dt), "psms", int(c.sum()))
out = np.zeros(32, np.uint64); L.check(L.load().sage_hip_debug_phase_cycles(scorer._h, L.as_ptr(out, C.c_uint64)))
ph = out[16:24].astype(np.float64)
nq = max(1, 2 * (t["n_wide"] + t["n_retry"]))  # items of both calls (pass 1 + retry pass)
print("count kernel, wave 0, cycles per queued spectrum: " + " ".join("%s=%d" % (k, v / nq) for k, v in
      zip(("query", "apply", "barrierA", "publish+load", "scan", "clear", "barrierC", "wait for cells"), ph)), " total=%d" % (ph.sum() / nq))
print("narrow prelim phases:", (out[0:8] / max(1, 2 * batch.n)).astype(int), " rescore phases:", (out[8:16] / max(1, 2 * batch.n)).astype(int))
r = out[8:16].astype(np.float64) / max(1, batch.n + t["n_retry"])
print("rescore per spectrum: load=%d score=%d sort=%d feature=%d cycles; items=%.1f longest=%.1f candidates=%.1f" % (r[0], r[1], r[3], r[4], r[5], r[6], r[7]))
