"""What does a spectrum cost as a function of its precursor mass?  (GPU probe, not a pytest.)

    python scripts/shard_cost_probe.py [C3] [slices=32] [steps=30]

The configuration's run cut into `slices` mass-contiguous pieces of EQUAL spectrum counts; per piece the resident step (ms) and
the sums of the candidate cost features over its spectra: n, peaks x fragment charges (the windows the preliminary kernel looks
up), peaks x fragment charges x candidates in the precursor window, min(candidates, 50) x precursor mass x fragment charges (ions
the rescoring kernel walks).  sage_amd/sharding.estimate_work's coefficients are the non-negative least-squares fit of the ms
column (profiles/r05_shard_cost_fit.txt)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import time  # noqa: E402

import bench  # noqa: E402
from sage_amd.api import DeviceDatabase, Scorer  # noqa: E402
from sage_amd.sharding import cost_features, plan_mass_shards, precursor_sort_mass  # noqa: E402
from sage_amd.workloads import CONFIGS, build_host_db, scorer_params  # noqa: E402

cfg_name = sys.argv[1] if len(sys.argv) > 1 else "C3"
slices = int(sys.argv[2]) if len(sys.argv) > 2 else 32
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
cfg = CONFIGS[cfg_name]
params = scorer_params(cfg)
host = build_host_db(cfg, peptides_only=True)
path = f"/tmp/ab_multi_{cfg_name}_{cfg['spectra']}.npz"
if os.path.exists(path):
    batch_all = bench.load_batch(path)
else:
    batch_all, _ = bench.generate_workload(cfg, host, cfg["spectra"])
    bench.save_batch(path, batch_all)
dev = DeviceDatabase(host, 0, build_on_device=True)
scorer = Scorer(dev, params)
feat = cost_features(batch_all.peak_off, batch_all.precursor_mz, batch_all.precursor_charge, params, host.pep_mono,
                     batch_all.isolation_lo, batch_all.isolation_hi)
mass = precursor_sort_mass(batch_all.precursor_mz, batch_all.precursor_charge, params)
print("FEATURES", " ".join(feat.keys()))
for k, idx in enumerate(plan_mass_shards(mass, slices, None)):
    b = batch_all.subset(idx)
    db = scorer.upload(b)
    for _ in range(3):
        scorer.score_resident(db)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            scorer.score_resident(db)
        best = min(best, (time.perf_counter() - t0) * 1e3 / steps)
    t = scorer.last_timing()
    print(f"SLICE {k} n {b.n} mass {np.nanmin(mass[idx]):.1f} {np.nanmax(mass[idx]):.1f} ms {best:.4f} prelim {t['prelim_ms']:.3f} rescore "
          f"{t['rescore_ms']:.3f} n_wide {t['n_wide']} ways {t['n_ways']} F " + " ".join(f"{feat[f][idx].sum():.6g}" for f in feat), flush=True)
    db.close()
