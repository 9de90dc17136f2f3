"""Ad-hoc GPU A/B (not a pytest): the resident scoring step of a BASELINE configuration under several environment settings,
one workload, one device database, a new scorer handle per setting (the knobs of DESIGN.md §8a are read when a handle is
created).  Prints ms per step (wall, HIP-event phases) per setting and checks that every setting returns the same PSMs.

usage: python scripts/ab_env.py C3 [n_spectra] -- "" "SAGE_HIP_XCD_CHUNK=0" "SAGE_HIP_XCD_CHUNK=256,SAGE_HIP_DEBUG_FLAGS=32"
       SAGE_HIP_LIB=<other .so> selects another build for the whole run (scripts/variants.sh)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402  (generate_workload: forked generation of the whole run)
from sage_amd.api import DeviceDatabase, Scorer  # noqa: E402
from sage_amd.workloads import CONFIGS, build_host_db, scorer_params  # noqa: E402

args = sys.argv[1:]
sets = [""]
if "--" in args:
    i = args.index("--")
    args, sets = args[:i], args[i + 1:]
name = args[0] if args else "C3"
cfg = CONFIGS[name]
n = int(args[1]) if len(args) > 1 else cfg["spectra"]
steps = int(os.environ.get("AB_STEPS", "10"))
params = scorer_params(cfg)
if os.environ.get("AB_REPORT_PSMS"):  # e.g. 50: the kernels for lists wider than a wavefront
    from dataclasses import replace
    params = replace(params, report_psms=int(os.environ["AB_REPORT_PSMS"]))
host = build_host_db(cfg, peptides_only=True)
batch, _ = bench.generate_workload(cfg, host, n)
if os.environ.get("AB_SORT"):  # what a mass-ordered copy of the batch in HBM would buy: hand the batch over sorted already
    z = np.where(batch.precursor_charge == 0, 2, batch.precursor_charge).astype(np.float32)
    batch = batch.subset(np.argsort((batch.precursor_mz - np.float32(1.0072764)) * z, kind="stable"))
dev = DeviceDatabase(host, 0, build_on_device=True)
ref = None
for s in sets:
    kv = dict(a.split("=", 1) for a in s.split(",") if a)
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update(kv)
    try:
        scorer = Scorer(dev, params)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    db = scorer.upload(batch)
    for _ in range(2):
        f, c = scorer.score_resident(db)
    f, c = f.copy(), c.copy()
    pm, rm = [], []
    t0 = time.perf_counter()
    for _ in range(steps):
        scorer.score_resident(db)
        t = scorer.last_timing()
        pm.append(t["prelim_ms"])
        rm.append(t["rescore_ms"])
    ms = (time.perf_counter() - t0) * 1e3 / steps
    same = ""
    if ref is None:
        ref = (f, c)
    else:
        same = "same PSMs" if bench.same_psms(f, c, ref[0], ref[1]) else "DIFFERENT PSMs"
    h2h = ""
    if os.environ.get("AB_H2H"):  # host arrays in, host records out (sage_hip_score_batch), page-locked
        locked = batch.page_locked()
        scorer.score(locked)
        t0 = time.perf_counter()
        for _ in range(steps):
            scorer.score(locked)
        h2h = f"  host-to-host {batch.n * steps / (time.perf_counter() - t0) / 1e6:.2f} M spectra/s"
        del locked
    print(f"{name} n={batch.n} [{s or 'default'}]:{h2h} {ms:.3f} ms/step  {batch.n / ms / 1e3:.2f} M spectra/s  prelim {np.mean(pm):.3f}  "
          f"rescore {np.mean(rm):.3f}  retry pass {t['retry_ms']:.3f}  wide {t['n_wide']} retry {t['n_retry']} tied {t['n_tied']} ways {t['n_ways']} wall {t['total_ms']:.3f}  psms {int(c.sum())}  {same}", flush=True)
    scorer.close()
