"""Multi-GPU sharding of a spectrum batch: one process per GPU, index replicated, spectra partitioned.

Each MS2 spectrum is scored independently against a read-only index (sage-cli runner.rs:311-325), so the
path shards with NO data-path collective.  The only communication is the host-side gather of the (small)
PSM records in input order — the reference's `collect()` preserves input order (runner.rs:325).

Two plans: `plan_shards` (contiguous in input position, balanced by work) and — the default of bench.py's strong scaling and of
cli.py --devices since round 5 — `plan_mass_shards` (contiguous in precursor mass, balanced by work): a rank then walks 1 / N of
the mass-sorted index instead of all of it.
"""
import numpy as np


PROTON = np.float32(1.0072764)
NEUTRON = np.float32(1.00335)


def estimate_work(peak_off, precursor_mz, precursor_charge, params, pep_mono, isolation_lo=None, isolation_hi=None):
    """Per-spectrum work estimate  P x sum over queries of (W + W0)  (peaks x queries x window size):
    P = peaks, one query per (precursor charge, isotope error) the scorer will evaluate (scoring.rs:384-462), W = candidates in
    the query's precursor window — two binary searches over the mass-sorted peptide list per query, IndexedDatabase::query's own
    lookup (database.rs:402-425) — and W0 = 64 for the per-query cost that does not depend on the window.  On a real run precursor
    mass, hence W, drifts with retention time, so contiguous shards of equal peak counts are not shards of equal work; in an open
    search W spans orders of magnitude.  float64 arithmetic: an estimate, not the kernels' window."""
    pep_mono = np.asarray(pep_mono, dtype=np.float64)
    n = len(peak_off) - 1
    peaks = np.diff(np.asarray(peak_off).astype(np.int64)).astype(np.float64)
    mz = np.asarray(precursor_mz, dtype=np.float64) - float(PROTON)
    z_in = np.asarray(precursor_charge).astype(np.int64)
    ranged = params.wide_window or params.override_precursor_charge
    isos = range(params.min_isotope_err, params.max_isotope_err + 1)
    work = np.zeros(n)
    for z in range(1, 256):
        use = (z_in == z) & (not ranged)
        if params.min_precursor_charge <= z <= params.max_precursor_charge:
            use = use | (z_in == 0) | ranged
        if not np.any(use):
            continue
        mass = mz[use] * z
        if params.wide_window:  # the isolation window scaled by the charge (scoring.rs:437-441), +-2.4 Th when absent
            lo_t = np.full(mass.shape, -2.4) if isolation_lo is None else np.where(np.isnan(isolation_lo[use]), -2.4, isolation_lo[use])
            hi_t = np.full(mass.shape, 2.4) if isolation_hi is None else np.where(np.isnan(isolation_hi[use]), 2.4, isolation_hi[use])
            lo_of = lambda c: c + lo_t * z  # noqa: E731
            hi_of = lambda c: c + hi_t * z  # noqa: E731
        else:
            t = params.precursor_tol
            if t.kind == "ppm":
                lo_of = lambda c: c * (1.0 + t.lo * 1e-6)  # noqa: E731
                hi_of = lambda c: c * (1.0 + t.hi * 1e-6)  # noqa: E731
            elif t.kind == "pct":
                lo_of = lambda c: c * (1.0 + t.lo * 1e-2)  # noqa: E731
                hi_of = lambda c: c * (1.0 + t.hi * 1e-2)  # noqa: E731
            else:
                lo_of = lambda c: c + t.lo  # noqa: E731
                hi_of = lambda c: c + t.hi  # noqa: E731
        acc = np.zeros(mass.shape)
        for iso in isos:
            c = mass - iso * float(NEUTRON)
            w = np.searchsorted(pep_mono, hi_of(c), side="right") - np.searchsorted(pep_mono, lo_of(c), side="left")
            acc += np.maximum(w, 0) + 64.0
        work[use] += acc
    return (peaks + 1.0) * np.maximum(work, 64.0)


def cost_features(peak_off, precursor_mz, precursor_charge, params, pep_mono, isolation_lo=None, isolation_hi=None):
    """Per-spectrum quantities the two narrow-search kernels' times follow (scripts/shard_cost_probe.py fits their weights):
    `n` 1; `windows` peaks x fragment charges (the (peak, charge) windows the preliminary kernel looks up, scoring.rs:358-375);
    `window_cands` windows x candidates in the precursor window (index entries it walks); `ions` min(candidates, 50) x precursor
    mass / 100 x fragment charges (the ion tables the rescoring kernel walks for the k-selected candidates: a peptide has ~ mass /
    110 residues).  Known charges only matter here; unknown charges take the scorer's range (an estimate, not the kernels' values)."""
    n = len(peak_off) - 1
    peaks = np.diff(np.asarray(peak_off).astype(np.int64)).astype(np.float64)
    z = np.asarray(precursor_charge).astype(np.int64)
    ranged = params.wide_window or params.override_precursor_charge
    zq = np.where((z == 0) | ranged, params.max_precursor_charge, z)
    user = -1 if params.max_fragment_charge is None else int(params.max_fragment_charge)
    inner = np.where(user >= 0, (user + 1) & 0xFF, zq)
    nfz = np.maximum(np.minimum(zq, inner), 2) - 1  # scoring.rs:239-247, exclusive bound - 1
    base = estimate_work(peak_off, precursor_mz, precursor_charge, params, pep_mono, isolation_lo, isolation_hi)
    cands = np.maximum(base / (peaks + 1.0) - 64.0, 0.0)  # (sum over the queries of the window sizes)
    mass = (np.asarray(precursor_mz, dtype=np.float64) - float(PROTON)) * np.maximum(zq, 1)
    return {"n": np.ones(n), "windows": peaks * nfz, "window_cands": peaks * nfz * cands,
            "ions": np.minimum(cands, 50.0) * np.nan_to_num(mass) / 100.0 * nfz}


def plan_shards(peak_off: np.ndarray, world: int, weights=None):
    """Contiguous, work-balanced shards: split the spectrum list where the cumulative work crosses k/world of the total.
    `weights`: per-spectrum work (estimate_work: peaks x queries x candidates in the precursor window); without it the peak
    count + 1 stands in (enough for a shuffled narrow search, where every window holds about as many candidates).
    Returns [(begin, end)] * world."""
    n = len(peak_off) - 1
    if world <= 1 or n == 0:
        return [(0, n)] + [(n, n)] * (max(world, 1) - 1)
    if weights is None:
        cum = peak_off[1:].astype(np.float64) + np.arange(1, n + 1)  # +1 per spectrum: empty spectra still cost a launch slot
    else:
        w = np.asarray(weights, dtype=np.float64)
        assert len(w) == n and np.all(w >= 0)
        cum = np.cumsum(np.maximum(w, 1e-9))
    total = cum[-1]
    cuts = [0]
    for k in range(1, world):
        cuts.append(int(np.searchsorted(cum, total * k / world, side="left")))
    cuts.append(n)
    cuts = np.maximum.accumulate(np.array(cuts))
    return [(int(cuts[i]), int(cuts[i + 1])) for i in range(world)]


def precursor_sort_mass(precursor_mz, precursor_charge, params):
    """The mass a spectrum is placed at on the index's mass axis: (mz - proton) x charge for an annotated charge, x the smallest
    charge the scorer would try otherwise (scoring.rs:384-462).  Only a sort key — which rank scores a spectrum changes
    nothing about its result."""
    z = np.asarray(precursor_charge).astype(np.float64)
    ranged = params.wide_window or params.override_precursor_charge
    z = np.where((z == 0) | ranged, float(params.min_precursor_charge), z)
    return (np.asarray(precursor_mz, dtype=np.float64) - float(PROTON)) * z


def plan_mass_shards(sort_mass, world: int, weights=None, blocks_per_rank: int = 16, light_refine: int = 1):
    """Shards made of runs that are contiguous in PRECURSOR MASS, not in input position (VERDICT r04 task 2): the spectra are
    ordered by mass (stable: input position breaks ties), the ordered list is cut into world x blocks_per_rank blocks of equal
    cumulative work, and rank r scores one block of every stride of `world` blocks (the r-th, and in every other stride the r-th
    from the end).  The peptide list is mass sorted and a precursor
    window is a run of it (database.rs:402-425), so inside each of its blocks a rank sees the same spectrum density per Dalton as
    a single GPU scoring the whole batch — neighbouring spectra share position-table rows, fragment tiles and ion tables in
    cache — and it touches ~1 / world of every per-peptide structure of the replicated index; input-contiguous shards see the
    whole index at 1 / world of the density.  Why blocks and not ONE mass range per rank: what a spectrum costs changes with its
    mass in ways the work estimate does not capture (measured on C3, profiles/r05_shard_sizes.txt: equal estimated work put
    65 000 spectra into the lightest octile and 82 000 into the heaviest, 0.79 / 0.54 / 1.05 ms per step for octiles 0 / 1 / 7);
    a rank that takes every world-th block samples the whole mass axis, so the shards balance whatever the cost profile, and a
    block of 1 / 64 of a run is still tens of times larger than the kernels' reuse distance.  blocks_per_rank=1 gives one range
    per rank; light_refine: see below.  Returns `world` index arrays (global input positions, ascending inside a shard, so a shard's records keep their
    relative input order); they partition range(n)."""
    m = np.asarray(sort_mass, dtype=np.float64)
    n = len(m)
    if world <= 1 or n == 0:
        return [np.arange(n, dtype=np.int64)] + [np.zeros(0, dtype=np.int64)] * (max(world, 1) - 1)
    order = np.argsort(np.where(np.isnan(m), np.inf, m), kind="stable")
    w = np.ones(n) if weights is None else np.maximum(np.asarray(weights, dtype=np.float64), 1e-9)
    assert len(w) == n
    cum = np.cumsum(w[order])
    nb = world * max(1, int(blocks_per_rank))
    # light_refine > 1 (NOT the default since round 6): the lightest two strides of blocks cut that many times finer.  Round 5 fitted it
    # on C3 — what a spectrum costs changes fastest at the light end of the axis, and the slowest of eight shards went from 0.659 to
    # 0.649 ms per step with 8 — and made it the default; held against workloads it was not fitted on (profiles/r06_shard_sizes.txt:
    # slowest / mean shard of eight, plain | refined: C3 1.034 | 1.020, the tie-rich C3T 1.010 | 1.014, C2 within the noise of a
    # 0.15 ms step) it buys a per cent on the workload it came from and nothing elsewhere: a constant of one benchmark, so the plan
    # is back to equal blocks.  The snake order below and "every rank samples the whole axis" are what balance the shards.
    fr = [k / nb for k in range(1, nb)]
    if light_refine > 1 and blocks_per_rank >= 4:
        # (cut positions as integers over the common denominator nb * light_refine: no two equal floats to deduplicate, and the
        # block count stays a multiple of `world` by construction — ADVICE r05)
        den = nb * light_refine
        nums = sorted(set(range(1, 2 * world * light_refine)) | set(k * light_refine for k in range(2 * world, nb)))
        fr = [k / den for k in nums]
        nb = len(fr) + 1
        assert nb % world == 0, (nb, world)
    cuts = [0] + [int(np.searchsorted(cum, cum[-1] * f, side="left")) for f in fr] + [n]
    cuts = np.maximum.accumulate(np.array(cuts))
    # boustrophedon: in every other stride of `world` blocks the ranks take their block in reverse order — a rank that always took
    # the r-th block of a stride would be systematically lighter than rank r + 1 (measured on C3 with 8 x 8 blocks: 0.629 ms for
    # rank 0 against 0.673 ms for rank 7 per 62 500 spectra)
    def blocks_of(r):
        return [j * world + (r if j % 2 == 0 else world - 1 - r) for j in range(nb // world)]
    return [np.sort(np.concatenate([order[cuts[b]:cuts[b + 1]] for b in blocks_of(r)])).astype(np.int64) for r in range(world)]


class FileExchange:
    """all_gather of Python objects between the ranks of ONE node through files in a directory they share, with nothing but a
    barrier from torch.distributed (bench.py hands its RCCL barrier in).  torch's all_gather_object pickles into device tensors and
    runs two collectives per call; this is the plumbing-proof alternative for the few host-side exchanges of a multi-GPU run (the
    mass plan before the timed region, the ordered gather of the records after it — `runner.rs:325`'s collect()): an exchange that
    cannot fail on a communicator's object path.  Every rank writes `<tag>_<rank>.pkl` (atomically: rename), all wait, every rank
    reads all `world` files, all wait, every rank removes its own."""

    def __init__(self, directory: str, rank: int, world: int, barrier):
        self.dir, self.rank, self.world, self.barrier = directory, int(rank), int(world), barrier
        self.calls = 0
        self.ranks_seen = 0  # files read by the last all_gather (== world, or the call raised)
        self.seconds = 0.0   # wall time spent in all_gather calls so far

    def all_gather(self, obj, tag: str = "x"):
        import os
        import pickle
        import time
        t0 = time.perf_counter()
        name = f"{tag}{self.calls}"
        self.calls += 1
        mine = os.path.join(self.dir, f"{name}_{self.rank}.pkl")
        with open(mine + ".tmp", "wb") as fh:
            pickle.dump(obj, fh, protocol=pickle.HIGHEST_PROTOCOL)
        os.replace(mine + ".tmp", mine)
        self.barrier()
        out = []
        for r in range(self.world):
            with open(os.path.join(self.dir, f"{name}_{r}.pkl"), "rb") as fh:
                out.append(pickle.load(fh))
        self.ranks_seen = len(out)
        self.barrier()
        try:
            os.unlink(mine)
        except OSError:
            pass
        self.seconds += time.perf_counter() - t0
        return out


def _all_gather(obj, group, exchange):
    """torch.distributed.all_gather_object, or the node-local file exchange when one is handed in"""
    if exchange is not None:
        return exchange.all_gather(obj, "gather")
    import torch.distributed as dist
    parts = [None] * dist.get_world_size(group)
    dist.all_gather_object(parts, obj, group=group)
    return parts


def gather_features_by_index(feats: np.ndarray, counts: np.ndarray, index: np.ndarray, n_total: int, group=None, exchange=None):
    """The gather for shards that are NOT contiguous in the input (plan_mass_shards): rank r holds (features[n_r, report],
    counts[n_r]) of the spectra at global input positions index[n_r]; every rank gets the whole result in input order — the
    reference's `collect()` order (runner.rs:325) — with spec_index rebased to the global batch.  A permutation on the host,
    no reduction; torch.distributed (RCCL on GPUs, gloo on CPU) — or `exchange`, a FileExchange — only carries the records."""
    f = feats.copy()
    index = np.asarray(index, dtype=np.int64)
    assert len(index) == len(counts) == f.shape[0]
    valid = np.arange(f.shape[1])[None, :] < counts[:, None]
    f["spec_index"] = np.where(valid, index[:, None].astype(np.uint32), f["spec_index"])
    parts = _all_gather((index, f, counts), group, exchange)
    out_f = np.zeros((n_total, f.shape[1]), dtype=f.dtype)
    out_c = np.zeros(n_total, dtype=counts.dtype)
    seen = np.zeros(n_total, dtype=bool)
    for idx, pf, pc in parts:
        assert not seen[idx].any(), "two ranks scored the same spectrum"
        seen[idx] = True
        out_f[idx], out_c[idx] = pf, pc
    assert seen.all(), "a spectrum was scored by no rank"
    return out_f, out_c


def shard_indices(peak_off: np.ndarray, rank: int, world: int) -> np.ndarray:
    b, e = plan_shards(peak_off, world)[rank]
    return np.arange(b, e)


def gather_features(feats: np.ndarray, counts: np.ndarray, begin: int, group=None, exchange=None):
    """Gather per-rank (features[n_r, report], counts[n_r]) to every rank, concatenated in input order, with
    spec_index rebased to the global batch.  Uses torch.distributed (RCCL on GPUs, gloo on CPU) or `exchange` (a FileExchange);
    host-side only."""
    f = feats.copy()
    valid = np.arange(f.shape[1])[None, :] < counts[:, None]  # slots beyond counts[i] stay zeroed
    f["spec_index"][valid] += np.uint32(begin)
    parts = _all_gather((begin, f, counts), group, exchange)
    parts.sort(key=lambda p: p[0])
    return np.concatenate([p[1] for p in parts], axis=0), np.concatenate([p[2] for p in parts], axis=0)
