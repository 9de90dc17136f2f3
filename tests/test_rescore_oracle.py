"""The rescoring oracle (oracle/rescore_oracle.cpp) against the reference's known-answer test and against independent
numpy restatements of the pieces the reference does not pin — CPU only."""
import numpy as np
import pytest

import oracle_lib
from rescore_utils import synthetic_features
from sage_amd.api import DatabaseParameters, Tolerance


def test_reference_kat_linear_discriminant():
    """crates/sage/src/ml/linear_discriminant.rs:238-288: 8 rows, 4 features, normalised projections to 1e-8."""
    feats = np.array([[5, 4, 3, 2], [4, 5, 4, 3], [6, 3, 4, 5], [1, 0, 2, 9], [5, 4, 4, 3], [2, 1, 1, 9.5], [1, 0, 2, 8],
                      [3, 2, -2, 10]], dtype=np.float64)
    decoy = np.array([0, 0, 0, 1, 0, 1, 1, 1], dtype=np.uint8)
    coef = oracle_lib.lda_train(feats, decoy)
    assert coef is not None
    scores = np.array([sum(w * x for w, x in zip(coef, row)) for row in feats])
    scores /= np.sqrt(np.sum(scores ** 2))
    expected = [0.49706043, 0.48920177, 0.48920177, -0.07209359, 0.51204672, -0.02849527, -0.04924864, -0.06055943]
    assert np.all(np.abs(scores - expected) <= 1e-8)


def test_lda_needs_both_classes():
    rows = np.random.default_rng(0).normal(size=(10, 3))
    assert oracle_lib.lda_train(rows, np.zeros(10, np.uint8)) is None
    assert oracle_lib.lda_train(rows, np.ones(10, np.uint8)) is None


def test_gauss_solve_with_epsilon_ladder():
    rng = np.random.default_rng(1)
    a = rng.normal(size=(6, 6))
    spd = a @ a.T + 6 * np.eye(6)
    b = rng.normal(size=6)
    x = oracle_lib.gauss_solve(spd, b)
    assert np.allclose(x, np.linalg.solve(spd + 1e-8 * np.eye(6), b), rtol=1e-9, atol=1e-12)  # fill_zero(1e-8), gauss.rs:62
    # a zero row/column (a constant feature): the regularised system still solves, that coefficient is 0 / eps = 0
    spd[3, :] = 0
    spd[:, 3] = 0
    b[3] = 0
    x = oracle_lib.gauss_solve(spd, b)
    assert x is not None and x[3] == 0.0


def _numpy_kde(scores, decoys, monotonic, bins, bw_mult):
    """kde.rs:21-133 restated with numpy (independent of the C++ oracle)."""
    d, t = scores[decoys == 1], scores[decoys == 0]
    pi = len(d) / len(scores)

    def pdf(sample, x):
        sigma = np.sqrt(np.mean((sample - np.mean(sample)) ** 2))
        h = sigma * (4.0 / 3.0 / len(sample)) ** 0.2 * bw_mult
        return np.exp(-0.5 * ((x[:, None] - sample[None, :]) / h) ** 2).sum(axis=1) / (np.sqrt(2 * np.pi) * h * len(sample))

    lo, hi = scores.min(), scores.max()
    step = (hi - lo) / (bins - 1)
    x = np.arange(bins) * step + lo
    dd, tt = pdf(d, x) * pi, pdf(t, x) * (1 - pi)
    pep = dd / (tt + dd)
    if monotonic:
        pep = np.maximum.accumulate(pep[::-1])[::-1]
    return pep, lo, step


@pytest.mark.parametrize("monotonic,bins,bw", [(True, 1000, 1.0), (False, 100, 2.0)])
def test_kde_against_numpy(monotonic, bins, bw):
    rng = np.random.default_rng(2)
    decoys = (rng.random(3000) < 0.4).astype(np.uint8)
    scores = np.where(decoys == 1, rng.normal(-1, 1, 3000), rng.normal(2, 1.5, 3000))
    q = rng.uniform(scores.min() - 1, scores.max() + 1, 50)
    ob, lo, step, pep = oracle_lib.kde(scores, decoys, monotonic, bins, bw, q)
    nb, nlo, nstep = _numpy_kde(scores, decoys, monotonic, bins, bw)
    assert lo == nlo and step == nstep
    assert np.allclose(ob, nb, rtol=1e-10, atol=1e-300)
    # Estimator::posterior_error (kde.rs:146-168): linear interpolation, clamped bin index, extrapolation below/above
    last = bins - 1
    for x, p in zip(q, pep):
        b = int(min(last, max(0, np.floor((x - lo) / step))))
        h = min(last, b + 1)
        assert np.isclose(p, ob[b] + (ob[h] - ob[b]) * ((x - (b * step + lo)) / step), rtol=1e-12, atol=1e-300)


def _numpy_q(score_sorted_decoy):
    d = 1 + np.cumsum(score_sorted_decoy)
    t = np.cumsum(~score_sorted_decoy)
    with np.errstate(divide="ignore"):
        q = d.astype(np.float32) / t.astype(np.float32)
    return np.minimum(np.minimum.accumulate(q[::-1])[::-1], np.float32(1.0))


@pytest.mark.parametrize("det", [False, True])
def test_rescore_pipeline_pieces(det):
    """det=False: the reference's order + platform libm; det=True: the device's arithmetic contract (detmath.h)."""
    f, pk, npk, prk, npr = synthetic_features(4000, seed=3)
    tol = Tolerance("ppm", -10.0, 10.0)
    r = oracle_lib.rescore(f, tol, pk, npk, prk, npr, want_rows=True, det=det)
    assert r["lda_fitted"]
    decoy = f["label"] == -1
    # the design: spot-check columns of compute_features (linear_discriminant.rs:162-195)
    rows = r["rows"]
    assert np.allclose(rows[:, 2], np.log1p(f["hyperscore"]))
    assert np.allclose(rows[:, 8], np.log1p(-f["poisson"]))
    assert np.all(rows[:, 10] == f["matched_peaks"]) and np.allclose(rows[:, 13], f["longest_y"] / f["peptide_len"])
    assert np.allclose(rows[:, 18], np.sqrt(0.999)) and np.all(rows[:, 16] == f["rt"].astype(np.float64))
    # coefficients: LDA direction solves Sw w = mu_t - mu_d (regularised)
    mu_t, mu_d = rows[~decoy].mean(0), rows[decoy].mean(0)
    sw = np.cov(rows[~decoy].T, bias=True) + np.cov(rows[decoy].T, bias=True)
    w = np.linalg.solve(sw + 1e-8 * np.eye(20), mu_t - mu_d)
    # (columns 18 / 19 are constant here: their coefficients are rounding noise times 1 / eps, different in every summation order)
    assert np.allclose(r["coef"][:18], w[:18], rtol=1e-5, atol=1e-6) and np.all(np.abs(r["coef"][18:]) < 1e-4)
    disc = rows @ r["coef"]
    assert np.allclose(r["discriminant_score"], disc.astype(np.float32), rtol=1e-6)
    # targets separate from decoys
    assert np.median(disc[~decoy]) > np.median(disc[decoy])
    # order: stable, descending total order; q-values from the counts
    order = r["order"]
    assert np.all(np.diff(r["discriminant_score"][order]) <= 0)
    q = _numpy_q(decoy[order])
    assert np.array_equal(r["spectrum_q"][order], q)
    assert r["passing"][0] == np.sum(q <= 0.01)
    assert np.all(r["posterior_error"] <= 0.0 + 1e-6) and np.all(r["posterior_error"] >= -324.0)
    # picked peptide: per (key, side) maxima, one q per side, features of a side share it; shared proteins keep 1.0
    for k in np.unique(pk)[:50]:
        for side in (False, True):
            m = (pk == k) & (decoy == side)
            if m.any():
                assert len(np.unique(r["peptide_q"][m])) == 1
    assert np.all(r["protein_q"][prk == 0xFFFFFFFF] == 1.0)
    assert np.all((r["peptide_q"] > 0) & (r["peptide_q"] <= 1.0))


def test_picked_competition_worked_example():
    """fdr.rs:60-120 by hand on 3 keys: rows sorted by score, decoy += pep, target += 1, reverse cumulative minimum."""
    f, *_ = synthetic_features(6, seed=4)
    f["label"] = [1, -1, 1, 1, -1, 1]
    f["poisson"] = [-9.0, -1.0, -7.0, -3.0, -2.0, -8.0]
    f["longest_y_pct"] = 0.0
    pk = np.array([0, 0, 1, 2, 2, 0], dtype=np.uint32)
    prk = np.full(6, 0xFFFFFFFF, dtype=np.uint32)
    r = oracle_lib.rescore(f[:1].repeat(6) if False else f, Tolerance("ppm", -10.0, 10.0), pk, 3, prk, 0)
    disc = r["discriminant_score"]
    decoy = f["label"] == -1
    # rows (key, side) -> best score
    rows = {}
    for i in range(6):
        key = (int(pk[i]), bool(decoy[i]))
        rows[key] = max(rows.get(key, -np.inf), float(disc[i]))
    win_s = np.array([max(rows.get((k, False), np.finfo(np.float32).min), rows.get((k, True), np.finfo(np.float32).min)) for k in range(3)])
    win_d = np.array([rows.get((k, True), np.finfo(np.float32).min) >= rows.get((k, False), np.finfo(np.float32).min) for k in range(3)])
    srt = sorted(rows.items(), key=lambda kv: -kv[1])
    _, _, _, pep = oracle_lib.kde(win_s, win_d.astype(np.uint8), True, 1000, 1.0, [s for _, s in srt])
    d, t, qs = np.float32(1.0), np.float32(0.0), []
    for (key, s), p in zip(srt, pep):
        d = np.float32(d + np.float32(p))
        if not key[1]:
            t = np.float32(t + 1)
        with np.errstate(divide="ignore"):
            qs.append(np.float32(d) / np.float32(t))
    qmin, out = np.float32(1.0), {}
    for (key, _), q in zip(reversed(srt), reversed(qs)):
        qmin = min(qmin, q) if q == q else qmin
        out[key] = qmin
    for i in range(6):
        exp = out[(int(pk[i]), bool(decoy[i]))]
        got = r["peptide_q"][i]
        assert (np.isnan(exp) and np.isnan(got)) or got == exp


def test_competition_keys_pair_targets_with_their_decoys():
    fasta = ">sp|P1|A\nMKAAAGGGLLLKDDDEEEFFFKWWWYYYR\n>sp|P2|B\nMKAAAGGGLLLKCCCHHHIIIK\n"
    params = DatabaseParameters.from_json({"enzyme": {"missed_cleavages": 0, "min_len": 5}, "static_mods": {}, "decoy_tag": "rev_",
                                           "generate_decoys": True, "fasta": "x"})
    host = params.build(fasta, peptides_only=True)
    n = host.n_peptides
    pk, npk, prk, npr = host.competition_keys(np.arange(n, dtype=np.uint32))
    strings = [host.peptide_string(i) for i in range(n)]
    decoy = host.decoy.astype(bool)
    assert decoy.any() and (~decoy).any()

    def flip(s):
        return s if len(s) < 3 else s[0] + s[1:-1][::-1] + s[-1]

    for i in range(n):
        key_string = flip(strings[i]) if decoy[i] else strings[i]
        for j in range(n):
            other = flip(strings[j]) if decoy[j] else strings[j]
            assert (pk[i] == pk[j]) == (key_string == other)
    assert npk == len(set(pk)) and set(pk) == set(range(npk))
    # AAAGGGLLLK occurs in both proteins: shared -> no protein key; unique peptides get their protein's key
    for i in range(n):
        nprot, _ = host.peptide_info(i)
        assert (prk[i] == 0xFFFFFFFF) == (nprot != 1)
    assert npr == len(set(prk[prk != 0xFFFFFFFF]))


def test_heuristic_fallback_when_the_pivot_search_skips_a_column():
    """A constant ims column gives an exactly-zero scatter row; with this data the signed-maximum pivot search of
    gauss.rs:97-108 lands on that zero and skips the column, left_solved() fails for every epsilon, and spectrum_fdr falls
    back to ln_1p(-poisson) + longest_y_pct / 3 (runner.rs:285-288) with posterior_error left at its default."""
    f, pk, npk, prk, npr = synthetic_features(4000, seed=3, zero_ims=True)
    r = oracle_lib.rescore(f, Tolerance("ppm", -10.0, 10.0), pk, npk, prk, npr)
    assert not r["lda_fitted"]
    exp = np.log1p((-f["poisson"]).astype(np.float32)) + f["longest_y_pct"] / np.float32(3.0)
    assert np.allclose(r["discriminant_score"], exp, rtol=1e-6)
    assert np.all(r["posterior_error"] == 1.0)


# ---- the predict_rt block (runner.rs:513-530) ------------------------------------------------------------------------------
def test_reference_kat_linear_regression():
    """crates/sage/src/ml/regression.rs:124-157"""
    x = np.arange(50, dtype=np.float64)
    beta, r2 = oracle_lib.linreg_fit(np.stack([x, np.ones(50)], 1), 2.0 * x + 1.0)
    assert abs(beta[0] - 2.0) < 1e-9 and abs(beta[1] - 1.0) < 1e-9 and abs(r2 - 1.0) < 1e-9  # fit_perfect_line
    i = np.arange(200, dtype=np.float64)
    x = i / 10.0
    beta, r2 = oracle_lib.linreg_fit(np.stack([x, np.ones(200)], 1), 3.0 * x + 2.0 + np.sin(i * 0.7) * 0.1)
    assert abs(beta[0] - 3.0) < 0.05 and abs(beta[1] - 2.0) < 0.1 and r2 > 0.99  # fit_with_noise
    assert oracle_lib.linreg_fit(np.ones((3, 1)), [1.0, 2.0, 3.0], keep=[0, 0, 0]) is None  # empty_filter_returns_none


def test_reference_kat_mobility_embedding():
    """crates/sage/src/ml/mobility_model.rs:188-267: N- / C-terminal residue counts of four peptides at charge 2"""
    valid = "ACDEFGHIKLMNPQRSTVWYUO"
    idx = {c: i for i, c in enumerate(valid)}
    n_term, c_term = 44, 66
    emb = [oracle_lib.im_embed(s, 1000.0, 2) for s in ("LEKSLIEK", "LERSLIEWK", "LWESLIEK", "CHADWICK")]
    assert [e[n_term + idx["L"]] for e in emb] == [1.0, 1.0, 1.0, 0.0]
    assert [e[n_term + idx["K"]] for e in emb] == [0.0, 0.0, 0.0, 0.0]
    assert [e[n_term + idx["W"]] for e in emb] == [0.0, 0.0, 1.0, 0.0]
    assert [e[c_term + idx["K"]] for e in emb] == [1.0, 1.0, 1.0, 1.0]
    assert [e[c_term + idx["W"]] for e in emb] == [0.0, 1.0, 0.0, 0.0]
    assert [e[c_term + idx["I"]] for e in emb] == [0.0, 0.0, 0.0, 0.0]
    e = emb[0]
    assert e[100 - 5] == 2.0 and e[100 - 6] == 0.5 and e[100 - 3] == 8.0 and e[100 - 1] == 1.0 and e[100 - 2] == 1.0
    assert np.isclose(e[22 + idx["L"]], 2 / 8) and e[idx["E"]] == 2.0
    # retention embedding (retention_model.rs:44-62): counts, first two residues, residues len-3 and len-2, len, ln1p(mass), 1
    r = oracle_lib.rt_embed("LEKSLIEK", 944.5)
    assert r[idx["L"]] == 2 and r[22 + idx["L"]] == 1 and r[22 + idx["E"]] == 1 and r[44 + idx["I"]] == 1 and r[44 + idx["E"]] == 1
    assert r[44 + idx["K"]] == 0 and r[66] == 8.0 and np.isclose(r[67], np.log1p(np.float32(944.5))) and r[68] == 1.0


def test_predict_rt_block_properties():
    """runner.rs:513-530 restated: poisson-sorted q-values, alignment, the two linear models — structural checks"""
    from sage_amd.synthetic import synthetic_rt_world
    f, off, seq, mono = synthetic_rt_world(12000, n_files=3, seed=21)
    r = oracle_lib.predict_rt(f, 3, off, seq, mono)
    decoy = f["label"] == -1
    # q-values of the poisson-sorted pass (ascending poisson = best first)
    order = np.argsort(f["poisson"], kind="stable")
    assert np.array_equal(r["spectrum_q"][order], _numpy_q(decoy[order]))
    train = (~decoy) & (r["spectrum_q"] <= 0.01)
    assert train.sum() > 1000
    # alignment: max_rt = ceil of the largest rt of the file; aligned_rt is the f32 formula of retention_alignment.rs:171
    al = r["alignments"]
    for k in range(3):
        assert al[k, 0] == np.ceil(f["rt"][f["file_id"] == k].max())
    a = al[f["file_id"]]
    assert np.array_equal(r["aligned_rt"], (f["rt"] / a[:, 0]) * a[:, 1] + a[:, 2])
    # the per-file regression maps each file onto the across-file mean: the same peptide seen in two files lands close
    assert np.all((al[:, 1] > 0.5) & (al[:, 1] < 1.5)) and np.all(np.abs(al[:, 2]) < 0.2)
    # models: fitted, predictions clamped, deltas as defined; independent least squares on the oracle's own embedding
    assert r["fitted"].all() and 0.2 < r["r2"][0] <= 1.0
    assert np.all((r["predicted_rt"] >= 0) & (r["predicted_rt"] <= 1)) and np.all((r["predicted_ims"] >= 0) & (r["predicted_ims"] <= 2))
    assert np.array_equal(r["delta_rt_model"], np.abs(r["aligned_rt"] - r["predicted_rt"]))
    assert np.array_equal(r["delta_ims_model"], np.abs(f["ims"] - r["predicted_ims"]))
    strs = [bytes(seq[int(off[i]):int(off[i + 1])]).decode() for i in range(len(f))]
    rows = np.stack([oracle_lib.rt_embed(strs[i], float(mono[i])) for i in np.flatnonzero(train)])
    y = r["aligned_rt"][train].astype(np.float64)
    beta = np.linalg.lstsq(rows, y, rcond=None)[0]
    pred = np.clip(rows @ beta, 0, 1)
    assert np.max(np.abs(pred - r["predicted_rt"][train])) < 5e-4  # (the reference solves the 1e-8-regularised normal equations)
    # true identifications are predicted better than random ones
    assert np.median(r["delta_rt_model"][train]) < 0.3 * np.median(r["delta_rt_model"][decoy])
    # one file, nothing to align against: slope ~ 1, intercept ~ 0
    f1 = f.copy()
    f1["file_id"] = 0
    r1 = oracle_lib.predict_rt(f1, 1, off, seq, mono)
    assert abs(r1["alignments"][0, 1] - 1.0) < 1e-5 and abs(r1["alignments"][0, 2]) < 1e-5
    # no training PSM at all: alignment falls back to slope 1 / intercept 0, models stay unfitted, defaults remain
    f0 = f.copy()
    f0["label"] = -1
    r0 = oracle_lib.predict_rt(f0, 3, off, seq, mono)
    assert not r0["fitted"].any() and np.all(r0["alignments"][:, 1] == 1.0) and np.all(r0["alignments"][:, 2] == 0.0)
    assert np.all(r0["delta_rt_model"] == np.float32(0.999)) and np.all(r0["predicted_rt"] == 0.0)


def test_fit_or_heuristic_decision_is_the_reference_orders():
    """Constant ims column (every search without ion mobility): whether the LDA is fitted or the heuristic discriminant is used
    hangs on the signs of rounding noise in the elimination (gauss.rs:97-108).  The LDA sums run in the reference's row order in
    both modes of the oracle (and on the device), so the contract's elementary functions — an ulp from the libm here and there —
    are all that could still tip it: they do not, on any of these data sets."""
    tol = Tolerance("ppm", -10.0, 10.0)
    seen = set()
    for n, seed in ((4000, 3), (900, 14), (12000, 15), (50000, 16), (2500, 17), (8000, 18), (20000, 19)):
        f, pk, npk, prk, npr = synthetic_features(n, seed=seed, zero_ims=True)
        a = oracle_lib.rescore(f, tol, pk, npk, prk, npr, det=False)
        b = oracle_lib.rescore(f, tol, pk, npk, prk, npr, det=True)
        assert a["lda_fitted"] == b["lda_fitted"], (n, seed)
        seen.add(bool(a["lda_fitted"]))
    assert seen == {True, False}  # (both branches occur in this set)


def test_det_contract_against_reference_order():
    """How far the arithmetic contract the device evaluates (sage_amd/csrc/detmath.h: IEEE-only ln_1p / exp, blocked order of
    the kernel-density sums; the LDA sums are sequential either way) is from the reference's own order with the platform libm:
    the design agrees to an ulp, coefficients to 1e-10, discriminants and q-values are equal."""
    tol = Tolerance("ppm", -10.0, 10.0)
    for n, seed in ((800, 21), (30000, 22)):
        f, pk, npk, prk, npr = synthetic_features(n, seed=seed)
        a = oracle_lib.rescore(f, tol, pk, npk, prk, npr, want_rows=True, det=False)
        b = oracle_lib.rescore(f, tol, pk, npk, prk, npr, want_rows=True, det=True)
        assert a["lda_fitted"] and b["lda_fitted"]
        # the design differs only where det_log1p / det_exp differ from the libm by an ulp
        assert np.allclose(a["rows"], b["rows"], rtol=1e-13, atol=1e-15)
        live = np.array([np.ptp(a["rows"][:, j]) > 0 for j in range(20)])
        scale = np.abs(a["coef"][live]).max()
        assert np.allclose(a["coef"][live], b["coef"][live], rtol=1e-10, atol=1e-10 * scale)
        assert np.allclose(a["discriminant_score"], b["discriminant_score"], rtol=1e-6, atol=1e-6)
        for k in ("spectrum_q", "peptide_q", "protein_q"):
            assert np.mean(a[k] == b[k]) > 0.9999, k
        assert abs(int(a["passing"][0]) - int(b["passing"][0])) <= 1


def test_det_math_against_libm():
    """det_log1p / det_exp (through the oracle's det mode on a one-column problem) stay within 1 ulp of numpy's libm."""
    x = np.concatenate([np.linspace(-0.99, 5.0, 20001), np.geomspace(1e-12, 1e6, 20001)])
    f, pk, npk, prk, npr = synthetic_features(len(x), seed=23)
    f["hyperscore"] = x
    r = oracle_lib.rescore(f, Tolerance("ppm", -10.0, 10.0), pk, npk, prk, npr, want_rows=True, det=True)
    got, exp = r["rows"][:, 2], np.log1p(x)
    assert np.all(np.abs(got - exp) <= np.spacing(np.abs(exp)))
