#!/usr/bin/env python3
"""bench.py — spectra/sec of the fragment-index search-and-score path on MI355X.

Default workload = BASELINE.json configs[2] ("C3", the configuration the metric is quoted on: human tryptic
narrow search): synthetic human-sized tryptic digest (20 400 proteins, 1 missed cleavage, static C+57.0215,
variable M+15.9949 and protein-N-term +42.0106, decoys on), ±10 ppm precursor and fragment tolerance,
report_psms 1, 62 500 synthetic MS2 spectra per GPU (= 500 000 at 8 GPUs).  --config C2 | C4 | C5 select the
other BASELINE.json configurations.  One "step" = Scorer::score over the whole resident batch (preliminary
fragment matching + k-select + rescoring + Feature assembly + D2H of the PSM records).  Spectra are sharded
across ranks, the index is replicated per GPU, no collective on the data path (weak scaling).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)

_ENZ1 = dict(missed_cleavages=1, min_len=5, max_len=50, cleave_at="KR", restrict="P")
_ENZ2 = dict(missed_cleavages=2, min_len=5, max_len=50, cleave_at="KR", restrict="P")
_DB = dict(bucket_size=8192, peptide_min_mass=500.0, peptide_max_mass=5000.0, static_mods={"C": 57.0215}, generate_decoys=True)

# BASELINE.json configs[1..4] (recipes: SURVEY.md §8d).  `spectra` is the size the config names; `per_gpu` is what ONE
# rank scores (weak scaling: at --gpus 8 the whole job is exactly the named size for C3/C4/C5).
CONFIGS = {
    "C2": dict(name="C2: 50k synthetic MS2 x yeast-like tryptic digest, ±10 ppm narrow search", proteins=6000,
               fasta_seed=1001, spectra=50000, per_gpu=50000, spectra_seed=2001, db=dict(_DB, enzyme=_ENZ1),
               scorer=dict(), spectra_kwargs=dict(), cpu_sample=50000,
               metric="spectra/sec (whole node), fragment-index search-and-score, narrow search"),
    "C3": dict(name="C3: 500k synthetic MS2 x human-like tryptic digest + 2 variable mods (M+15.9949, protein N-term "
                    "+42.0106), ±10 ppm narrow search, 62 500 spectra per GPU", proteins=20400,
               fasta_seed=1002, spectra=500000, per_gpu=62500, spectra_seed=2002,
               db=dict(_DB, enzyme=_ENZ1, variable_mods={"M": [15.9949], "[": [42.010565]}, max_variable_mods=2),
               scorer=dict(), spectra_kwargs=dict(varmod_frac=0.15), cpu_sample=62500,
               metric="spectra/sec (whole node), fragment-index search-and-score, human tryptic narrow search"),
    "C4": dict(name="C4: 100k synthetic MS2 x human-like tryptic digest (2 missed cleavages, variable M+15.9949), open "
                    "search da[-500,100], 12 500 spectra per GPU", proteins=20400,
               fasta_seed=1002, spectra=100000, per_gpu=12500, spectra_seed=2004,
               db=dict(_DB, enzyme=_ENZ2, variable_mods={"M": [15.9949]}, max_variable_mods=2),
               scorer=dict(precursor_tol=("da", -500.0, 100.0)), spectra_kwargs=dict(mass_shift_frac=0.3),
               cpu_sample=2048,
               metric="spectra/sec (whole node), fragment-index search-and-score, open search"),
    "C5": dict(name="C5: 200k chimeric synthetic MS2 (2-3 peptides per 12 Th isolation window, no charge annotation) x "
                    "human-like tryptic digest + 2 variable mods, wide_window + chimera, report_psms 5, 25 000 spectra per GPU",
               proteins=20400, fasta_seed=1002, spectra=200000, per_gpu=25000, spectra_seed=2005,
               db=dict(_DB, enzyme=_ENZ1, variable_mods={"M": [15.9949], "[": [42.010565]}, max_variable_mods=2),
               scorer=dict(wide_window=True, chimera=True, report_psms=5, min_precursor_charge=2, max_precursor_charge=4),
               spectra_kwargs=dict(chimeric=3, isolation_half_width=6.0, annotate_charge=False), cpu_sample=4096,
               metric="spectra/sec (whole node), fragment-index search-and-score, chimeric wide-window search"),
}
DEFAULT_CONFIG = "C3"  # BASELINE.json's metric is quoted on the human tryptic narrow search; it fits one GPU


def _scorer_params(cfg):
    from sage_amd.api import ScorerParams, Tolerance
    kw = dict(cfg["scorer"])
    for k in ("precursor_tol", "fragment_tol"):
        if k in kw:
            kw[k] = Tolerance(*kw[k])
    return ScorerParams(**kw)


def _tol_str(t):
    return f"{t.kind}[{t.lo:g},{t.hi:g}]"


def rescore_bench(args):
    """Secondary measurement: the post-search rescoring (mass-error KDE, LDA, posterior-error KDE, q-values, picked peptide /
    protein FDR) over one synthetic Feature table resident on the host, single GPU.  A "step" is one complete rescoring."""
    import numpy as np

    from sage_amd.api import Tolerance, device_count, rescore
    from sage_amd.synthetic import synthetic_features

    if device_count() <= 0:
        raise SystemExit("bench.py needs a HIP device (libsage_hip has no CPU fallback)")
    n = args.rescore_psms
    f, pk, npk, prk, npr = synthetic_features(n, seed=77)
    tol = Tolerance("ppm", -10.0, 10.0)
    for _ in range(max(1, args.warmup)):
        res = rescore(f, tol, pk, npk, prk, npr)
    wall, dev = [], []
    for _ in range(args.steps):
        t0 = time.perf_counter()
        res = rescore(f, tol, pk, npk, prk, npr)
        wall.append((time.perf_counter() - t0) * 1e3)
        dev.append(res.device_ms)
    ms = float(np.mean(wall))
    # the Gaussian-kernel sums dominate: (mass bins + 1000 + 1000 + 1000) bins x samples evaluations of exp()
    kde_evals = 100 * n + 1000 * n + 1000 * npk + 1000 * npr
    cpu = None
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib  # the CPU restatement: checker and baseline, never the product
        m = min(n, args.cpu_sample or 100_000)
        idx = np.arange(m)
        pk_s = np.unique(pk[idx], return_inverse=True)[1].astype(np.uint32)
        sel = prk[idx] != 0xFFFFFFFF
        prk_s = np.full(m, 0xFFFFFFFF, dtype=np.uint32)
        prk_s[sel] = np.unique(prk[idx][sel], return_inverse=True)[1].astype(np.uint32)
        t0 = time.perf_counter()
        o = oracle_lib.rescore(f[idx], tol, pk_s, int(pk_s.max()) + 1, prk_s, int(prk_s[sel].max()) + 1 if sel.any() else 0)
        t_cpu = time.perf_counter() - t0
        g = rescore(f[idx], tol, pk_s, int(pk_s.max()) + 1, prk_s, int(prk_s[sel].max()) + 1 if sel.any() else 0)
        same = float(np.mean(np.isclose(g.spectrum_q, o["spectrum_q"], rtol=1e-4)))
        cpu = {"value": m / t_cpu, "unit": "PSMs/s", "cores": 1, "kind": "port",
               "sample": f"the first {m} PSMs, one pass, sequential restatement (the reference parallelises the KDE sums with rayon)",
               "parity": f"lda_fitted {g.lda_fitted}=={o['lda_fitted']}, spectrum_q equal on {same:.4%} of PSMs, "
                         f"passing {g.passing_spectrum} vs {int(o['passing'][0])}"}
    print(json.dumps({
        "metric": "PSMs/sec, post-search rescoring (LDA + KDE posterior error + q-values + picked FDR)", "value": n * 1e3 / ms,
        "unit": "PSMs/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{n} synthetic PSMs (35 % decoys), ppm[-10,10], {npk} peptide keys, {npr} protein keys",
                   "lda_fitted": bool(res.lda_fitted), "passing_spectrum": res.passing_spectrum},
        "device_ms": float(np.mean(dev)), "kde_exp_evaluations": kde_evals,
        "kde_gexp_per_s": kde_evals / (float(np.mean(dev)) * 1e-3) / 1e9, "cpu_baseline": cpu}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default=DEFAULT_CONFIG, choices=sorted(CONFIGS))
    ap.add_argument("--spectra", type=int, default=0, help="override the number of spectra per rank (smoke runs)")
    ap.add_argument("--proteins", type=int, default=0, help="override the number of proteins (smoke runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-repeats", type=int, default=3)
    ap.add_argument("--cpu-sample", type=int, default=0, help="override the number of spectra the CPU baseline scores")
    ap.add_argument("--rescore-psms", type=int, default=0,
                    help="measure the post-search rescoring (sage_hip_rescore, SURVEY 8f rank 4) on this many synthetic PSMs "
                         "instead of the search path; not the headline metric")
    args = ap.parse_args()
    if args.rescore_psms:
        return rescore_bench(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import numpy as np
    import torch  # plumbing only: device sync + the inter-rank barrier

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (libsage_hip has no CPU fallback)")
    # SAGE_BENCH_BACKEND=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (ranks share devices, the
    # barrier / max-over-ranks run over gloo on CPU tensors); the driver's runs use nccl (= RCCL), one rank per GPU
    backend = os.environ.get("SAGE_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    coll_device = "cuda" if backend == "nccl" else "cpu"

    from sage_amd.api import DatabaseParameters, DeviceDatabase, Scorer, SpectrumBatch, SpectrumProcessor
    from sage_amd.synthetic import synthetic_fasta, synthetic_spectra

    cfg = CONFIGS[args.config]
    n_prot = args.proteins or cfg["proteins"]
    n_spec = args.spectra or cfg["per_gpu"]
    full_size = not args.spectra and not args.proteins
    t0 = time.time()
    fasta = synthetic_fasta(n_prot, cfg["fasta_seed"])
    # the reference-shaped fragment arrays are only needed by the CPU oracle leg (rank 0 of a 1-GPU run); the GPU index is
    # generated on the device from the peptide list either way
    need_oracle = world == 1 and not args.no_cpu_baseline
    host = DatabaseParameters(**cfg["db"]).build(fasta, peptides_only=not need_oracle)
    t_db = time.time() - t0
    t0 = time.time()
    raw = synthetic_spectra(host, n_spec, cfg["spectra_seed"] + rank, **cfg["spectra_kwargs"])
    sp = SpectrumProcessor(150, True, 0.0)  # max_peaks 150, deisotope (input.rs:366, 371)
    proc = [sp.process(r) for r in raw]
    proc = [p for p in proc if len(p.masses) >= 15]  # min_peaks 15 (runner.rs:313)
    batch = SpectrumBatch.from_spectra(proc)
    t_spec = time.time() - t0
    del raw, proc

    params = _scorer_params(cfg)
    t0 = time.time()
    # the fragment index is generated on the device from the peptide list (index_build.hip); the host-built fragments above
    # only feed the CPU oracle leg
    dev = DeviceDatabase(host, local_rank, build_on_device=True)
    t_dev = time.time() - t0
    scorer = Scorer(dev, params)
    dbatch = scorer.upload(batch)  # inputs resident in HBM before the timed region

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        feats, counts = scorer.score_resident(dbatch)
    prelim_ms, rescore_ms = [], []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        feats, counts = scorer.score_resident(dbatch)
        t = scorer.last_timing()
        prelim_ms.append(t["prelim_ms"])
        rescore_ms.append(t["rescore_ms"])
    barrier()
    elapsed = time.perf_counter() - t0
    last_t = scorer.last_timing()
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        nn = torch.tensor([batch.n], dtype=torch.int64, device=coll_device)
        dist.all_reduce(nn, op=dist.ReduceOp.SUM)
        total_spectra = int(nn.item())
    else:
        total_spectra = batch.n
    ms_per_step = elapsed * 1000.0 / args.steps
    value = total_spectra * args.steps / elapsed

    if rank == 0:
        n_psm = int(counts.sum())
        feats, counts = feats.copy(), counts.copy()  # the pinned result buffers are reused by the next call
        # PCIe-inclusive rate (host buffers in, host records out) — reported beside `value`, never as `value`
        t0 = time.perf_counter()
        for _ in range(3):
            scorer.score(batch)
        pcie_value = batch.n * 3 / (time.perf_counter() - t0)
        # ---- cpu_baseline + algorithmic bytes: the oracle (restated reference CPU path), rank 0, N=1 only
        cpu = None
        bytes_per_spec = None
        work = None
        cache = os.path.join(ROOT, "profiles", "algorithmic_bytes.json")
        cached = json.load(open(cache)) if os.path.exists(cache) else {}
        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib
            from parity_utils import assert_features_equal
            orc = oracle_lib.OracleDb.from_product(host)
            threads = os.cpu_count() or 1
            n_cpu = min(batch.n, args.cpu_sample or cfg["cpu_sample"])
            sample = batch if n_cpu == batch.n else batch.subset(np.arange(n_cpu))
            runs = []
            for r in range(args.cpu_repeats + 1):  # first run = warm-up (and the work counters)
                of, oc, ms, wk = orc.score(params, sample, threads=threads, work=(r == 0))
                if r:
                    runs.append(sample.n * 1000.0 / (ms + 1.0))  # runner.rs:327-330
                else:
                    work = wk
            parity_psms = assert_features_equal(feats[:n_cpu], counts[:n_cpu], of, oc, "bench parity")  # same inputs
            what = f"all {batch.n} spectra of the workload" if n_cpu == batch.n else \
                f"the first {n_cpu} of the {batch.n} spectra of the workload"
            cpu = {"value": float(np.median(runs)), "unit": "spectra/s", "cores": threads, "kind": "port",
                   "sample": f"{what}, median of {args.cpu_repeats} passes after 1 warm-up, {threads} OpenMP threads, "
                             f"dynamic schedule (restated reference CPU path, not Sage itself)",
                   "parity": f"{parity_psms} PSMs identical to the GPU result (ints/f32 exact, f64 within 1e-12)"}
            rescore_bytes = 4 * work["rescored"] + 5 * work["rescored_residues"] + 64 * work["reported"]
            bytes_per_spec = {"total": work["algorithmic_bytes"] / sample.n,
                              "prelim": (work["algorithmic_bytes"] - rescore_bytes) / sample.n,
                              "rescore": rescore_bytes / sample.n}
            if full_size:
                try:
                    cached[args.config] = {"bytes_per_spectrum": bytes_per_spec, "work": work, "n_spectra": sample.n}
                    os.makedirs(os.path.dirname(cache), exist_ok=True)
                    json.dump(cached, open(cache, "w"), indent=1)
                except OSError:
                    pass
        elif args.config in cached:
            bytes_per_spec = cached[args.config]["bytes_per_spectrum"]

        pm, rm = float(np.mean(prelim_ms)), float(np.mean(rescore_ms))
        dom = "prelim" if pm >= rm else "rescore"
        roof = None
        if bytes_per_spec:
            dom_ms = max(pm, rm)
            achieved = bytes_per_spec[dom] * batch.n / (dom_ms * 1e-3) / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath) and full_size:
                traffic = json.load(open(tpath)).get(args.config, {}).get(dom + "_bytes_per_launch")
            roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "kernel_ms": {"prelim": pm, "rescore": rm},
                    "algorithmic_bytes_per_spectrum": bytes_per_spec,
                    "whole_path_achieved_GBs": bytes_per_spec["total"] * batch.n / ((pm + rm) * 1e-3) / 1e9,
                    "routing": {"spectra": batch.n, "large_window_kernel": last_t["n_wide"],
                                "exact_retry_for_tied_hyperscores": last_t["n_retry"]},
                    "note": "prelim = fragment matching + k-select kernels (HIP events on the scorer's stream); narrow-window "
                            "searches are probe/latency bound: few algorithmic bytes per spectrum by construction"}
        out = {
            "metric": cfg["metric"],
            "value": value, "unit": "spectra/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg["name"], "spectra_per_gpu": batch.n, "peptides": host.n_peptides,
                       "fragments": host.n_fragments if host.has_fragments else None, "precursor_tol": _tol_str(params.precursor_tol),
                       "fragment_tol": _tol_str(params.fragment_tol), "report_psms": params.report_psms,
                       "chimera": params.chimera, "wide_window": params.wide_window,
                       "parallelism": f"spectra sharded x{world}, index replicated",
                       "psms_per_step_rank0": n_psm,
                       "setup_s": {("db_build_host_incl_fragments_for_the_oracle" if need_oracle else "db_build_host_peptides"): round(t_db, 2),
                                   "spectra": round(t_spec, 2),
                                   "index_build_on_device": round(t_dev, 2)},
                       "index_device_bytes": dev.device_bytes},
            "roofline": roof, "cpu_baseline": cpu,
            "pcie_inclusive_value": pcie_value,
        }
        if cpu:
            out["speedup_vs_cpu_baseline"] = value / cpu["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
