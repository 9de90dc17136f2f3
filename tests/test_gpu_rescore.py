"""GPU parity of the post-search rescoring (rescore.hip, through sage_hip_rescore) against the CPU oracle.

Reductions on the device are tree sums, the oracle's are left-to-right (the reference's own KDE sums run in rayon's
arbitrary order), so f64 intermediates agree to rounding, not bit for bit:
  coefficients 1e-6 relative (the regularised 20x20 system has condition number ~1e9), discriminant 1e-5, log10 posterior
  error 2e-3 absolute, q-values equal except where two PSMs are closer than that noise (at most 0.1 % of rows may differ).
"""
import numpy as np
import pytest

import oracle_lib
from rescore_utils import synthetic_features
from sage_amd.api import (DatabaseParameters, DeviceDatabase, Scorer, ScorerParams, SpectrumBatch, SpectrumProcessor, Tolerance,
                          rescore)
from sage_amd.synthetic import synthetic_fasta, synthetic_spectra

pytestmark = pytest.mark.gpu


def compare(f, tol, pk, npk, prk, npr, context, **opt):
    g = rescore(f, tol, pk, npk, prk, npr, **opt)
    o = oracle_lib.rescore(f, tol, pk, npk, prk, npr, want_rows=True, **opt)
    n = len(f)
    assert g.lda_fitted == o["lda_fitted"], context
    if g.lda_fitted:
        # constant columns (ims without ion mobility, the two model deltas at their 0.999 default) have a scatter row of
        # rounding noise over the 1e-8 regulariser: their coefficients are noise / 1e-8 on both sides, and multiply a constant
        live = np.array([np.ptp(o["rows"][:, j]) > 0 for j in range(20)]) if o.get("rows") is not None else np.ones(20, bool)
        scale = np.abs(o["coef"][live]).max()
        assert np.allclose(g.coef[live], o["coef"][live], rtol=1e-6, atol=1e-7 * scale), (context, g.coef, o["coef"])
        assert np.all(np.abs(g.coef[~live]) < 1e-2 * scale), (context, g.coef)
    # (the noise coefficients of constant columns shift every discriminant by the same constant: no effect on order, PEP, q)
    offset = float(np.median(g.discriminant_score.astype(np.float64) - o["discriminant_score"]))
    assert abs(offset) < 1e-3, (context, offset)
    assert np.allclose(g.discriminant_score - offset, o["discriminant_score"], rtol=1e-5, atol=1e-5), context
    bad = np.abs(g.posterior_error - o["posterior_error"]) > 2e-3
    assert bad.mean() <= 1e-3, (context, int(bad.sum()), g.posterior_error[bad][:5], o["posterior_error"][bad][:5])
    for name, got, exp in (("spectrum_q", g.spectrum_q, o["spectrum_q"]), ("peptide_q", g.peptide_q, o["peptide_q"]),
                           ("protein_q", g.protein_q, o["protein_q"])):
        both_nan = np.isnan(got) & np.isnan(exp)
        close = both_nan | np.isclose(got, exp, rtol=1e-4, atol=1e-7)
        assert (~close).mean() <= 1e-3, (context, name, int((~close).sum()), got[~close][:5], exp[~close][:5])
    # the output order: a permutation, descending, and the same as the oracle's wherever scores are not near-tied
    assert sorted(g.order.tolist()) == list(range(n)), context
    assert np.all(np.diff(g.discriminant_score[g.order]) <= 0), context
    assert (g.order != o["order"]).mean() <= 1e-2, context
    for got, exp in zip((g.passing_spectrum, g.passing_peptide, g.passing_protein), o["passing"]):
        assert abs(int(got) - int(exp)) <= max(2, int(exp) // 500), (context, got, exp)
    return g, o


@pytest.mark.parametrize("n,seed", [(20000, 5), (300, 6), (100000, 8)])
def test_rescore_synthetic_ppm(gpu_required, n, seed):
    f, pk, npk, prk, npr = synthetic_features(n, seed=seed)
    g, o = compare(f, Tolerance("ppm", -10.0, 10.0), pk, npk, prk, npr, f"ppm n={n}")
    assert g.lda_fitted and g.passing_spectrum > 0


def test_rescore_da_tolerance_and_model_inputs(gpu_required):
    """Da precursor tolerance: mass error = expmass - calcmass, 0.1 x bandwidth, max(hi - lo, 1000) bins
    (linear_discriminant.rs:140-157); aligned_rt / delta_rt_model / delta_ims_model arrays as the RT / IM models leave them."""
    f, pk, npk, prk, npr = synthetic_features(30000, seed=9, ppm=False)
    rng = np.random.default_rng(10)
    opt = dict(aligned_rt=rng.uniform(0, 1, len(f)).astype(np.float32),
               delta_rt_model=np.abs(rng.normal(0, 0.05, len(f))).astype(np.float32),
               delta_ims_model=np.abs(rng.normal(0, 0.02, len(f))).astype(np.float32))
    g, _ = compare(f, Tolerance("da", -500.0, 100.0), pk, npk, prk, npr, "da", **opt)
    assert g.lda_fitted


def test_rescore_fallback_paths(gpu_required):
    # no decoys at all: train returns None (linear_discriminant.rs:83-85) -> heuristic discriminant, q = 1 / targets
    f, pk, npk, prk, npr = synthetic_features(5000, seed=11, decoy_frac=0.0)
    g, o = compare(f, Tolerance("ppm", -10.0, 10.0), pk, npk, prk, npr, "no decoys")
    assert not g.lda_fitted and np.all(g.posterior_error == 1.0)
    # constant ims column: whether the reference's solve succeeds depends on rounding noise in the last pivots (see
    # tests/test_rescore_oracle.py); parity is required only when both sides take the same branch
    f, pk, npk, prk, npr = synthetic_features(4000, seed=3, zero_ims=True)
    tol = Tolerance("ppm", -10.0, 10.0)
    g = rescore(f, tol, pk, npk, prk, npr)
    o = oracle_lib.rescore(f, tol, pk, npk, prk, npr)
    if g.lda_fitted == o["lda_fitted"]:
        compare(f, tol, pk, npk, prk, npr, "zero ims")


def test_rescore_rejects_sparse_keys_and_pct(gpu_required):
    from sage_amd._lib import SageHipError
    f, pk, npk, prk, npr = synthetic_features(1000, seed=12)
    with pytest.raises(SageHipError):
        rescore(f, Tolerance("ppm", -10.0, 10.0), pk, npk + 5, prk, npr)
    with pytest.raises(SageHipError):
        rescore(f, Tolerance("pct", -1.0, 1.0), pk, npk, prk, npr)
    g = rescore(f[:0], Tolerance("ppm", -10.0, 10.0), pk[:0], 0, prk[:0], 0)
    assert g.passing_spectrum == 0 and not g.lda_fitted


def test_search_then_rescore_end_to_end(gpu_required):
    """Features straight out of Scorer::score, keys from the host database, through the rescoring — the runner.rs flow."""
    fasta = synthetic_fasta(400, seed=31)
    params = DatabaseParameters(bucket_size=2048, enzyme=dict(missed_cleavages=1, cleave_at="KR", restrict="P"),
                                static_mods={"C": 57.0215})
    host = params.build(fasta)
    dev = DeviceDatabase(host, 0)
    raw = synthetic_spectra(host, 3000, 32)
    sp = SpectrumProcessor(150, True, 0.0)
    batch = SpectrumBatch.from_spectra([sp.process(r) for r in raw])
    tol = Tolerance("ppm", -20.0, 20.0)
    scorer = Scorer(dev, ScorerParams(precursor_tol=tol, fragment_tol=Tolerance("ppm", -10.0, 10.0), report_psms=2))
    feats, counts = scorer.score(batch)
    flat = np.concatenate([feats[i, :counts[i]] for i in range(len(counts))])
    assert len(flat) > 1000 and (flat["label"] == -1).any()
    pk, npk, prk, npr = host.competition_keys(flat["peptide_idx"])
    g, o = compare(flat, tol, pk, npk, prk, npr, "end to end")
    # (whether the linear model is fitted here is up to the reference's pivot search: ims is constant, see
    # tests/test_rescore_oracle.py — either way true matches are found at 1 %)
    assert g.passing_spectrum > 100


# ---- the predict_rt block (runner.rs:513-530) on the device --------------------------------------------------------------
def compare_rt(f, n_files, off, seq, mono, context):
    from sage_amd.api import predict_rt
    g = predict_rt(f, n_files, off, seq, mono)
    o = oracle_lib.predict_rt(f, n_files, off, seq, mono)
    assert np.array_equal(g.spectrum_q, o["spectrum_q"]), context  # integer counts, min: exact
    assert np.array_equal(g.alignments["max_rt"], o["alignments"][:, 0]), context
    assert np.allclose(g.alignments["slope"], o["alignments"][:, 1], rtol=1e-5, atol=1e-6), context
    assert np.allclose(g.alignments["intercept"], o["alignments"][:, 2], rtol=1e-5, atol=1e-6), context
    assert np.allclose(g.aligned_rt, o["aligned_rt"], rtol=1e-5, atol=1e-6), context
    assert (g.rt_fitted, g.ims_fitted) == tuple(o["fitted"]), context
    # the normal equations are rank deficient (counts sum to the length, ...) and solved through a 1e-8 regulariser:
    # coefficients are noisy, predictions much less so (a few 1e-5 on values of order 0.1 - 1 with a few thousand rows)
    for name, got, exp in (("predicted_rt", g.predicted_rt, o["predicted_rt"]), ("delta_rt_model", g.delta_rt_model, o["delta_rt_model"]),
                           ("predicted_ims", g.predicted_ims, o["predicted_ims"]), ("delta_ims_model", g.delta_ims_model, o["delta_ims_model"])):
        assert np.allclose(got, exp, rtol=1e-3, atol=3e-4), (context, name, np.abs(got - exp).max())
    for fitted, got, exp in ((g.rt_fitted, g.rt_r2, o["r2"][0]), (g.ims_fitted, g.ims_r2, o["r2"][1])):
        if fitted:  # (all-zero ion mobilities: y_var == 0, r^2 = 1 - 0 / 0 on both sides)
            assert (np.isnan(got) and np.isnan(exp)) or abs(got - exp) < 1e-5, (context, got, exp)
    return g, o


@pytest.mark.parametrize("n,n_files,seed", [(20000, 3, 9), (500, 1, 10), (60000, 8, 11)])
def test_predict_rt_synthetic(gpu_required, n, n_files, seed):
    from sage_amd.synthetic import synthetic_rt_world
    f, off, seq, mono = synthetic_rt_world(n, n_files, seed=seed)
    g, _ = compare_rt(f, n_files, off, seq, mono, f"rt n={n} files={n_files}")
    assert g.rt_fitted


def test_predict_rt_degenerate_inputs(gpu_required):
    from sage_amd._lib import SageHipError
    from sage_amd.api import predict_rt
    from sage_amd.synthetic import synthetic_rt_world
    f, off, seq, mono = synthetic_rt_world(4000, 2, seed=12, with_ims=False)  # no ion mobility: the IM model fits zeros
    compare_rt(f, 2, off, seq, mono, "no ims")
    f0 = f.copy()
    f0["label"] = -1  # nothing to train on
    g, _ = compare_rt(f0, 2, off, seq, mono, "no targets")
    assert not g.rt_fitted and np.all(g.delta_rt_model == np.float32(0.999))
    with pytest.raises(SageHipError):
        predict_rt(f, 1, off, seq, mono)  # file_id 1 with n_files 1


def test_search_predict_rescore_chain(gpu_required):
    """search -> predict_rt -> rescore with the model outputs as LDA features, against the oracle's chain"""
    from sage_amd.api import predict_rt
    fasta = synthetic_fasta(400, seed=41)
    params = DatabaseParameters(bucket_size=2048, enzyme=dict(missed_cleavages=1, cleave_at="KR", restrict="P"),
                                static_mods={"C": 57.0215})
    host = params.build(fasta)
    dev = DeviceDatabase(host, 0)
    sp = SpectrumProcessor(150, True, 0.0)
    tol = Tolerance("ppm", -20.0, 20.0)
    scorer = Scorer(dev, ScorerParams(precursor_tol=tol, fragment_tol=Tolerance("ppm", -10.0, 10.0), report_psms=2))
    flat = []
    for file_id in range(2):
        raw = synthetic_spectra(host, 2000, 42 + file_id)
        batch = SpectrumBatch.from_spectra([sp.process(r) for r in raw])
        feats, counts = scorer.score(batch)
        part = np.concatenate([feats[i, :counts[i]] for i in range(len(counts))])
        part["file_id"] = file_id
        # synthetic spectra carry no retention time: give every peptide a reproducible one, stretched per file
        rng = np.random.default_rng(7)
        rt_of_pep = rng.uniform(5, 50, host.n_peptides).astype(np.float32)
        part["rt"] = rt_of_pep[part["peptide_idx"]] * np.float32(1.0 + 0.1 * file_id) + np.float32(2.0 * file_id)
        # ... and an ion mobility, so that no LDA column is constant: with constant columns the reference's elimination pivots
        # on rounding noise and neither side's coefficients mean anything
        ims_of_pep = rng.uniform(0.7, 1.2, host.n_peptides).astype(np.float32)
        part["ims"] = ims_of_pep[part["peptide_idx"]] + np.float32(0.1) * part["charge"] + rng.normal(0, 0.01, len(part)).astype(np.float32)
        flat.append(part)
    flat = np.concatenate(flat)
    off, seq, mono = host.feature_peptides(flat["peptide_idx"])
    g, o = compare_rt(flat, 2, off, seq, mono, "chain")
    pk, npk, prk, npr = host.competition_keys(flat["peptide_idx"])
    opt = dict(aligned_rt=o["aligned_rt"], delta_rt_model=o["delta_rt_model"], delta_ims_model=o["delta_ims_model"])
    compare(flat, tol, pk, npk, prk, npr, "chain rescore", **opt)
