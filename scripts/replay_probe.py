"""Ad-hoc GPU probe (not a pytest): what the heap replay of the exact retry pass does per query on a BASELINE config
(wavefront-per-query kernel): cycles, stream words, offers that enter.  usage: python scripts/replay_probe.py [config] [n]"""
import os, sys, numpy as np, ctypes as C
os.environ["SAGE_HIP_PHASE_CLOCKS"] = "1"
os.environ["SAGE_HIP_DEBUG_FLAGS"] = "1024"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from sage_amd import _lib as L
from sage_amd.api import DeviceDatabase, Scorer
from sage_amd.workloads import CONFIGS, build_host_db, scorer_params
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
cfg = CONFIGS[name]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
host = build_host_db(cfg, peptides_only=True)
batch, _ = bench.generate_workload(cfg, host, n)
dev = DeviceDatabase(host, 0, build_on_device=True)
scorer = Scorer(dev, scorer_params(cfg)); db = scorer.upload(batch)
scorer.score_resident(db)
t = scorer.last_timing()
out = np.zeros(32, np.uint64); L.check(L.load().sage_hip_debug_phase_cycles(scorer._h, L.as_ptr(out, C.c_uint64)))
cyc, off, q, words = float(out[24]), float(out[25]), max(1.0, float(out[29])), float(out[30])
print(name, t)
print("replayed queries %d: cycles/query %.0f  stream words/query %.0f  offers/query %.0f  cycles/offer %.0f" % (q, cyc / q, words / q, off / q, cyc / max(1.0, off)))
