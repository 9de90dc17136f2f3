"""Ad-hoc GPU probe (not a pytest): large-window kernel phase cycles on a C2-sized database.
usage: python scripts/tile_probe.py [n_spectra] [mode: open|wide]"""
import os, sys, time, numpy as np, ctypes as C
os.environ["SAGE_HIP_PHASE_CLOCKS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from sage_amd import _lib as L
from sage_amd.api import *
from sage_amd.synthetic import *
nspec = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
mode = sys.argv[2] if len(sys.argv) > 2 else "open"
cfg = bench.CONFIGS["C2"]
host = DatabaseParameters(**cfg["db"]).build(synthetic_fasta(cfg["proteins"], cfg["fasta_seed"]))
sp = SpectrumProcessor(150, True, 0.0)
if mode == "open":
    raw = synthetic_spectra(host, nspec, 2004, mass_shift_frac=0.3)
    params = ScorerParams(precursor_tol=Tolerance("da", -500.0, 100.0))
else:
    raw = synthetic_spectra(host, nspec, 2005, chimeric=3, isolation_half_width=6.0, annotate_charge=False)
    params = ScorerParams(wide_window=True, chimera=True, report_psms=5)
batch = SpectrumBatch.from_spectra([p for p in (sp.process(r) for r in raw) if len(p.masses) >= 15])
dev = DeviceDatabase(host, 0)
scorer = Scorer(dev, params); db = scorer.upload(batch)
scorer.score_resident(db)
t0 = time.perf_counter(); f, c = scorer.score_resident(db); dt = time.perf_counter() - t0
print(mode, scorer.last_timing(), "spectra/s %.4g" % (batch.n / dt), "psms", int(c.sum()))
out = np.zeros(32, np.uint64); L.check(L.load().sage_hip_debug_phase_cycles(scorer._h, L.as_ptr(out, C.c_uint64)))
nn = 2.0 * batch.n
print("  count kernel cycles/spectrum (wave 0): query %d stream %d wait1 %d pass1 %d wait2+alloc %d pass2 %d wait3 %d" % tuple(out[16:23] / nn), "arena entries/spectrum %.0f" % (scorer.last_timing()["arena_entries"] / batch.n))
nb = 2.0 * ((batch.n * 1 + 63) // 64)
print("  replay kernel per wave: build %d replay %d cycles" % tuple(out[24:26] / nb))
if "--check" in sys.argv:
    import oracle_lib
    from parity_utils import assert_features_equal
    sub = batch.subset(np.arange(min(batch.n, 512)))
    of, oc, ms, _ = oracle_lib.OracleDb.from_product(host).score(params, sub, threads=0)
    print("parity PSMs", assert_features_equal(f[:sub.n], c[:sub.n], of, oc, "probe"))
