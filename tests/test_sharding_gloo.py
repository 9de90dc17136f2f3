"""N>1 path on CPU: world_size-2 gloo run of the shard -> score -> ordered-gather plumbing.  The scoring leg
is the oracle here (there is no GPU in the build container); on GPUs the same plumbing wraps Scorer.score."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, plan="input"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist
    import oracle_lib
    from sage_amd.api import DatabaseParameters, ScorerParams, SpectrumBatch, SpectrumProcessor
    from sage_amd.sharding import (estimate_work, gather_features, gather_features_by_index, plan_mass_shards, plan_shards,
                                   precursor_sort_mass)
    from sage_amd.synthetic import synthetic_fasta, synthetic_spectra
    dist.init_process_group("gloo", rank=rank, world_size=world)
    host = DatabaseParameters(static_mods={"C": 57.0215}).build(synthetic_fasta(60, seed=3))
    sp = SpectrumProcessor(150, True, 0.0)
    batch = SpectrumBatch.from_spectra([sp.process(r) for r in synthetic_spectra(host, 101, seed=4)])
    orc = oracle_lib.OracleDb.from_product(host)
    params = ScorerParams(report_psms=2)
    if plan == "mass":  # shards contiguous in precursor mass, balanced by estimate_work; the gather is a permutation
        w = estimate_work(batch.peak_off, batch.precursor_mz, batch.precursor_charge, params, host.pep_mono)
        idx = plan_mass_shards(precursor_sort_mass(batch.precursor_mz, batch.precursor_charge, params), world, w)[rank]
        b, e = int(idx.min()) if len(idx) else 0, int(idx.max()) + 1 if len(idx) else 0
        f, c, _, _ = orc.score(params, batch.subset(idx), threads=1)
        gf, gc = gather_features_by_index(f, c, idx, batch.n)
    else:
        b, e = plan_shards(batch.peak_off, world)[rank]
        shard = batch.subset(np.arange(b, e))
        f, c, _, _ = orc.score(params, shard, threads=1)
        gf, gc = gather_features(f, c, b)
    if rank == 0:
        ff, fc, _, _ = orc.score(params, batch, threads=1)
        q.put((np.array_equal(gc, fc), gf.tobytes() == ff.tobytes(), int(gc.sum()), (b, e)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,plan", [(2, "input"), (8, "input"), (2, "mass"), (8, "mass")])
def test_shard_and_gather_preserves_input_order(world, plan):
    """world_size 2 and 8 (the node the path is sharded over): every rank scores its shard — a plan_shards range of the input or
    a plan_mass_shards slice of the precursor-mass axis — and the gathered records equal a single pass over the whole batch byte
    for byte, in input order."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, plan)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    counts_equal, feats_equal, n_psm, rng = res
    assert counts_equal and feats_equal and n_psm > 50


def test_plan_shards_properties():
    from sage_amd.sharding import plan_shards
    rng = np.random.default_rng(0)
    for world in (1, 2, 3, 8):
        for n in (0, 1, 5, 1000):
            lens = rng.integers(0, 200, n)
            off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
            shards = plan_shards(off, world)
            assert len(shards) == world and shards[0][0] == 0 and shards[-1][1] == n
            for (a, b), (c, d) in zip(shards, shards[1:]):
                assert b == c and a <= b
            if n == 1000 and world > 1:
                work = [int(off[e] - off[b]) for b, e in shards]
                assert max(work) < 1.2 * (sum(work) / world) + 400


def test_shards_weighted_by_window_size():
    """shards balanced on peaks x queries x candidates-in-window.  A run whose precursor masses drift upwards (as they do
    with retention time) through a peptide list that gets denser with mass: equal peak counts are then not equal work, the weighted
    plan is; and the estimate's window counts are IndexedDatabase::query's (two binary searches over the mass-sorted list)."""
    from sage_amd.api import ScorerParams, Tolerance
    from sage_amd.sharding import estimate_work, plan_shards
    rng = np.random.default_rng(3)
    pep_mono = np.sort(rng.uniform(500.0, 5000.0, 200_000) ** 1.0).astype(np.float32)
    pep_mono = np.sort((500.0 + 4500.0 * rng.beta(4.0, 1.5, 200_000)).astype(np.float32))  # dense at high mass
    n = 6000
    z = rng.choice([2, 3], n).astype(np.uint8)
    mass = np.sort(rng.uniform(600.0, 4800.0, n))  # "retention-time" drift
    mz = (mass / z + 1.0072764).astype(np.float32)
    peaks = rng.integers(80, 150, n)
    off = np.concatenate([[0], np.cumsum(peaks)]).astype(np.uint64)
    for params in (ScorerParams(precursor_tol=Tolerance("da", -50.0, 50.0)),
                   ScorerParams(precursor_tol=Tolerance("ppm", -10.0, 10.0), min_isotope_err=-1, max_isotope_err=3)):
        w = estimate_work(off, mz, z, params, pep_mono)
        assert w.shape == (n,) and np.all(w > 0)
        # spot-check the window count of a few spectra against a direct count
        if params.precursor_tol.kind == "da":
            for i in (0, n // 2, n - 1):
                c = float(mz[i] - np.float32(1.0072764)) * int(z[i])
                direct = int(np.sum((pep_mono >= c - 50.0) & (pep_mono <= c + 50.0)))
                assert abs(w[i] / (peaks[i] + 1.0) - (direct + 64.0)) <= 2.0
        world = 8
        plain, weighted = plan_shards(off, world), plan_shards(off, world, w)
        load = lambda shards: np.array([w[b:e].sum() for b, e in shards])  # noqa: E731
        assert weighted[0][0] == 0 and weighted[-1][1] == n and all(b == c for (_, b), (c, _) in zip(weighted, weighted[1:]))
        assert load(weighted).max() / load(weighted).mean() < 1.05
        if params.precursor_tol.kind == "da":
            assert load(plain).max() / load(plain).mean() > 1.3  # what balancing on peak counts alone leaves on the table
    # unknown charges: one query per charge of the configured range
    w2 = estimate_work(off, mz, np.zeros(n, np.uint8), ScorerParams(min_precursor_charge=2, max_precursor_charge=4), pep_mono)
    assert np.all(w2 >= (peaks + 1.0) * 3 * 64.0)


def test_plan_mass_shards_properties():
    """A partition of the input into `world` shards of equal estimated work, each made of blocks_per_rank runs of the mass axis
    (rank r: blocks r, r + world, ...): every spectrum in exactly one shard, ascending input positions inside a shard, NaN masses
    (no precursor) at the heavy end; with one block per rank shard r's masses all <= shard r + 1's."""
    from sage_amd.sharding import plan_mass_shards
    rng = np.random.default_rng(5)
    for world in (1, 2, 3, 8):
        for n in (0, 1, 7, 5000):
            m = rng.uniform(500.0, 5000.0, n)
            if n > 10:
                m[::97] = np.nan
                m[5:9] = m[4]  # equal masses: input position breaks the tie
            w = rng.uniform(1.0, 50.0, n)
            key = np.where(np.isnan(m), np.inf, m)
            for bpr in (1, 8):
                shards = plan_mass_shards(m, world, w, blocks_per_rank=bpr)
                assert len(shards) == world
                assert sorted(np.concatenate(shards).tolist()) == list(range(n))
                for sh in shards:
                    assert np.all(np.diff(sh) > 0)
                if bpr == 1:
                    tops = [key[sh].max() for sh in shards if len(sh)]
                    bots = [key[sh].min() for sh in shards if len(sh)]
                    assert all(t <= b for t, b in zip(tops, bots[1:]))
                if n == 5000 and world > 1:
                    load = np.array([w[sh].sum() for sh in shards])
                    assert load.max() / load.mean() < 1.05
                    if bpr == 8 and world == 8:  # every rank samples the whole axis: a cost the weights do not know balances too
                        hidden = np.where(key < 1000.0, 3.0, 1.0) * np.where(key > 4000.0, 2.0, 1.0)
                        hl = np.array([hidden[sh].sum() for sh in shards])
                        assert hl.max() / hl.mean() < 1.15
                        # and a shard still consists of few dense runs of the mass order: 8 runs, not thousands
                        rank_of = np.empty(n, dtype=np.int64)
                        for r, sh in enumerate(shards):
                            rank_of[sh] = r
                        runs = 1 + int(np.count_nonzero(np.diff(rank_of[np.argsort(key, kind="stable")]) != 0))
                        assert runs <= 8 * 8 + 2 * 8 * 8  # (+ the lightest two strides cut eight times finer)


def _exchange_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist
    import bench
    from sage_amd.api import DatabaseParameters, ScorerParams, SpectrumBatch, SpectrumProcessor
    from sage_amd.sharding import plan_mass_shards, precursor_sort_mass
    from sage_amd.synthetic import synthetic_fasta, synthetic_spectra
    dist.init_process_group("gloo", rank=rank, world_size=world)
    host = DatabaseParameters(static_mods={"C": 57.0215}).build(synthetic_fasta(40, seed=5), peptides_only=True)
    sp = SpectrumProcessor(150, True, 0.0)
    whole = SpectrumBatch.from_spectra([sp.process(r) for r in synthetic_spectra(host, 203, seed=6)])  # (every rank: the same run)
    params = ScorerParams()
    lo, hi = whole.n * rank // world, whole.n * (rank + 1) // world  # ... of which a rank holds its contiguous span, as bench.py's ranks do
    # (as bench.py does it: a directory of the launch agreed on through one tensor broadcast, then files + barriers only)
    from sage_amd.sharding import FileExchange, gather_features_by_index
    xchg = FileExchange(bench.shared_directory(dist, rank, "cpu"), rank, world, dist.barrier)
    shard, index, n_total = bench.exchange_by_mass(whole.subset(np.arange(lo, hi)), params, host.pep_mono, rank, world, xchg)
    plan = plan_mass_shards(precursor_sort_mass(whole.precursor_mz, whole.precursor_charge, params), world)
    want = whole.subset(plan[rank])
    same = n_total == whole.n and np.array_equal(index, plan[rank]) and shard.n == want.n
    for k in bench._BATCH_FIELDS:
        a, b = getattr(shard, k), getattr(want, k)
        same = same and ((a is None and b is None) or (a is not None and b is not None and a.tobytes() == b.tobytes()))
    # ... and the ordered gather through the same exchange gives what torch's own object collective gives
    from sage_amd import _lib as L
    f = np.zeros((shard.n, 1), dtype=L.FEATURE_DTYPE)
    f["peptide_idx"][:, 0] = (index % 97).astype(np.uint32)
    c = (index % 2).astype(np.uint32)
    gf1, gc1 = gather_features_by_index(f, c, index, n_total, exchange=xchg)
    gf2, gc2 = gather_features_by_index(f, c, index, n_total)
    same = same and gf1.tobytes() == gf2.tobytes() and gc1.tobytes() == gc2.tobytes() and xchg.ranks_seen == world
    same = same and np.array_equal(gc1, (np.arange(n_total) % 2).astype(np.uint32))
    ok = xchg.all_gather(bool(same), "ok")
    if rank == 0:
        import shutil
        shutil.rmtree(xchg.dir, ignore_errors=True)
        q.put(all(ok) and len(ok) == world)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_bench_exchange_by_mass_hands_every_rank_its_planned_shard(world):
    """bench.py's strong-scaling set-up without a GPU: every rank holds a contiguous span of THE run, the ranks agree on
    sharding.plan_mass_shards from the exchanged masses and hand the spectra over through node-local files; each rank ends up with
    exactly the planned shard — the same spectra, peaks and all, that a cut of the whole run would give it."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res is True
