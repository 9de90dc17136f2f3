"""Ad-hoc GPU probe (not a pytest): open-search throughput on the C2 database, GPU kernels vs oracle."""
import os, sys, time, numpy as np, ctypes as C
os.environ["SAGE_HIP_PHASE_CLOCKS"]="1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench, oracle_lib
from parity_utils import assert_features_equal
from sage_amd.api import *
from sage_amd.synthetic import *
cfg = bench.CONFIGS["C2"]
nspec = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
host = DatabaseParameters(**cfg["db"]).build(synthetic_fasta(cfg["proteins"], cfg["fasta_seed"]))
sp = SpectrumProcessor(150, True, 0.0)
batch = SpectrumBatch.from_spectra([p for p in (sp.process(r) for r in synthetic_spectra(host, nspec, 2004, mass_shift_frac=0.3)) if len(p.masses) >= 15])
params = ScorerParams(precursor_tol=Tolerance("da", -500.0, 100.0))
dev = DeviceDatabase(host, 0)
for label, env in (("open kernel", {}), ("no offers", {"SAGE_HIP_DEBUG_FLAGS":"1"}), ("no atomics", {"SAGE_HIP_DEBUG_FLAGS":"2"}), ("no slot scan", {"SAGE_HIP_DEBUG_FLAGS":"4"}), ("nothing", {"SAGE_HIP_DEBUG_FLAGS":"7"})):
    os.environ.pop("SAGE_HIP_OPEN_THRESH", None); os.environ.pop("SAGE_HIP_DEBUG_FLAGS", None); os.environ.update(env)
    scorer = Scorer(dev, params); db = scorer.upload(batch)
    scorer.score_resident(db)
    t0 = time.perf_counter(); f, c = scorer.score_resident(db); dt = time.perf_counter() - t0
    print(label, scorer.last_timing(), "spectra/s", batch.n / dt, "psms", int(c.sum()))
    if not env: feats, counts = f.copy(), c.copy()
    from sage_amd import _lib as L
    out=np.zeros(16,np.uint64); L.check(L.load().sage_hip_debug_phase_cycles(scorer._h, L.as_ptr(out,C.c_uint64)))
    raw=np.zeros((4096,16),np.uint64); L.check(L.load().sage_hip_debug_phase_raw(scorer._h, L.as_ptr(raw,C.c_uint64), 4096))
    tot=raw[:,5:8].sum(axis=1)/2.0; print("  per-block total ticks: min %.3g median %.3g p90 %.3g p99 %.3g max %.3g" % (tot.min(), np.median(tot), np.percentile(tot,90), np.percentile(tot,99), tot.max()))
    nn=2*min(batch.n,4096)
    print("  open-kernel cycles/spectrum: query %d match %d trim %d" % tuple(out[5:8]//nn))
    scorer.close()
orc = oracle_lib.OracleDb.from_product(host)
of, oc, ms, work = orc.score(params, batch, threads=0, work=True)
print("oracle ms", ms, "spectra/s", batch.n * 1000 / (ms + 1), "alg bytes/spectrum", work["algorithmic_bytes"] / batch.n, work)
print("parity PSMs", assert_features_equal(feats, counts, of, oc, "open probe"))
