"""Ad-hoc GPU probe: per-phase shader cycles of the narrow preliminary kernel and the rescoring kernel.
usage: python scripts/phase_clocks.py [config] [n_spectra]"""
import os, sys, numpy as np, ctypes as C
os.environ["SAGE_HIP_PHASE_CLOCKS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sage_amd import _lib as L
from sage_amd.api import *
from sage_amd.synthetic import *
import bench
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
host = DatabaseParameters(**cfg["db"]).build(synthetic_fasta(cfg["proteins"], cfg["fasta_seed"]))
sp = SpectrumProcessor(150, True, 0.0)
batch = SpectrumBatch.from_spectra([p for p in (sp.process(r) for r in synthetic_spectra(host, n, cfg["spectra_seed"], **cfg["spectra_kwargs"])) if len(p.masses) >= 15])
scorer = Scorer(DeviceDatabase(host, 0), bench._scorer_params(cfg)); db = scorer.upload(batch)
for _ in range(3): scorer.score_resident(db)
out = np.zeros(32, np.uint64); L.check(L.load().sage_hip_debug_phase_cycles(scorer._h, L.as_ptr(out, C.c_uint64)))
nn = 3 * batch.n
print("prelim  cycles/spectrum: staging %d search %d match %d trim %d output %d" % tuple(out[:5] // nn))
print("prelim  per spectrum: offers %.1f  potential %.1f  queries %.2f" % tuple(out[5:8] / nn))
print("rescore cycles/spectrum: setup %d phaseA1(match) %d phaseB %d rank %d emit %d phaseA0(gather) %d" % tuple(out[8:14] // nn))
print(scorer.last_timing())
