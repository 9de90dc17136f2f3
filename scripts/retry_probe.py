"""Ad-hoc GPU probe (not a pytest): in-kernel phase cycles of the exact retry pass (narrow_kernel in exact mode) on a BASELINE
config.  usage: python scripts/retry_probe.py [config] [n_spectra]"""
import os, sys, numpy as np, ctypes as C
os.environ["SAGE_HIP_PHASE_CLOCKS"] = "1"
os.environ["SAGE_HIP_DEBUG_FLAGS"] = "512"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from sage_amd import _lib as L
from sage_amd.api import DeviceDatabase, Scorer
from sage_amd.workloads import CONFIGS, build_host_db, scorer_params
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = CONFIGS[name]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 62500
host = build_host_db(cfg, peptides_only=True)
batch, _ = bench.generate_workload(cfg, host, n)
dev = DeviceDatabase(host, 0, build_on_device=True)
scorer = Scorer(dev, scorer_params(cfg)); db = scorer.upload(batch)
scorer.score_resident(db)
t = scorer.last_timing()
out = np.zeros(32, np.uint64); L.check(L.load().sage_hip_debug_phase_cycles(scorer._h, L.as_ptr(out, C.c_uint64)))
nr = max(1, t["n_retry"])
print(name, t)
print("exact retry pass, cycles per retried spectrum (100 MHz clock64 ticks x ?):")
print("  preliminary phase: windows=%d query=%d matching=%d k-select=%d list=%d" % tuple(out[i] / nr for i in (0, 1, 2, 3, 4)))
print("  rescoring phase:   load=%d score=%d sort=%d feature=%d" % tuple(out[8 + i] / nr for i in (0, 1, 3, 4)))
print("  heap offers per spectrum: %.1f  candidates per query %.1f" % (out[5] / nr, out[6] / max(1, out[7])))
