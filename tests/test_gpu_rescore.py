"""GPU parity of the post-search rescoring (rescore.hip, through sage_hip_rescore) against the CPU oracle.

The LDA sums (class means, within-class scatter) run strictly in row order on the device, as in the reference; what the product
fixes beyond that is in sage_amd/csrc/detmath.h (IEEE-only ln_1p / exp, blocked order of the kernel-density sums, whose order
the reference leaves to rayon).  Two comparisons per data set:
  * against the oracle in the same mode (`det=True`): every bit that reaches the Gauss-Jordan pivot search is then the same on
    both sides, so the fit-or-heuristic decision, the coefficients, the discriminants, every q-value and the output order are
    held EQUAL — also on data with constant columns (ims == 0 is the normal case without ion mobility), where the elimination
    pivots on rounding noise.  The one exception is log10 of the posterior error (device libm vs glibc): 2 f32 ulps;
  * against the oracle in the reference's own mode (`det=False`: platform libm, every sum sequential): the same
    fit-or-heuristic decision on every data set, and — where no column is constant, i.e. where the coefficients mean
    anything — coefficients to 1e-9, discriminants and posterior errors to 1e-5, spectrum / peptide / protein q-values and the
    output order equal (>= 99.99 % / 99.9 % of the PSMs).  This leg is the INDEPENDENT check: oracle/detmath_oracle.h, which the
    `det=True` leg uses, is a copy of the product's detmath.h (same algorithms, namespace renamed), kept so that the checker does
    not include product headers — equality with it proves the device evaluates the stated contract, not that the contract is right.
"""
import numpy as np
import pytest

import oracle_lib
from rescore_utils import synthetic_features
from sage_amd.api import (DatabaseParameters, DeviceDatabase, Scorer, ScorerParams, SpectrumBatch, SpectrumProcessor, Tolerance,
                          rescore)
from sage_amd.synthetic import synthetic_fasta, synthetic_spectra

pytestmark = pytest.mark.gpu


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.array_equal(a, b) or bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def compare(f, tol, pk, npk, prk, npr, context, **opt):
    g = rescore(f, tol, pk, npk, prk, npr, **opt)
    o = oracle_lib.rescore(f, tol, pk, npk, prk, npr, det=True, **opt)
    n = len(f)
    assert g.lda_fitted == o["lda_fitted"], context
    if g.lda_fitted:
        assert _same(g.coef, o["coef"]), (context, g.coef, o["coef"])  # f64, bit for bit
    assert _same(g.discriminant_score, o["discriminant_score"]), (context, np.flatnonzero(g.discriminant_score != o["discriminant_score"])[:5])
    pe_g, pe_o = g.posterior_error.astype(np.float64), o["posterior_error"].astype(np.float64)
    assert np.all(np.abs(pe_g - pe_o) <= 2.5e-7 * np.maximum(np.abs(pe_o), 1e-30)), (context, np.abs(pe_g - pe_o).max())
    for name, got, exp in (("spectrum_q", g.spectrum_q, o["spectrum_q"]), ("peptide_q", g.peptide_q, o["peptide_q"]),
                           ("protein_q", g.protein_q, o["protein_q"])):
        assert _same(got, exp), (context, name, np.flatnonzero(got != exp)[:5])
    assert np.array_equal(g.order, o["order"]), context
    assert (int(g.passing_spectrum), int(g.passing_peptide), int(g.passing_protein)) == tuple(int(x) for x in o["passing"]), context
    # ... and against the reference's own arithmetic (platform libm, sequential sums)
    r = oracle_lib.rescore(f, tol, pk, npk, prk, npr, det=False, want_rows=True, **opt)
    assert g.lda_fitted == r["lda_fitted"], (context, "fit-or-heuristic decision differs from the reference order")
    if g.lda_fitted and all(np.ptp(r["rows"][:, j]) > 0 for j in range(r["rows"].shape[1])):
        scale = np.abs(r["coef"]).max()
        assert np.allclose(g.coef, r["coef"], rtol=1e-9, atol=1e-9 * scale), (context, np.abs(g.coef - r["coef"]).max() / scale)
        assert np.mean(g.spectrum_q == r["spectrum_q"]) > 0.9999, context
        # the per-PSM outputs, too: discriminants and posterior errors (f32 of f64 arithmetic that differs in the last bits between
        # the two libms and summation orders: measured equal on every data set of this file, held to 1e-5), and the q-values of the
        # two picked competitions (equal unless such a last-bit difference swaps two neighbours of the sort)
        assert np.allclose(g.discriminant_score, r["discriminant_score"], rtol=1e-5, atol=1e-6), (context, "discriminant_score vs the reference order")
        assert np.allclose(g.posterior_error, r["posterior_error"], rtol=1e-5, atol=1e-6), (context, "posterior_error vs the reference order")
        for name, got in (("peptide_q", g.peptide_q), ("protein_q", g.protein_q)):
            assert np.mean(got == r[name]) > 0.9999, (context, name, "vs the reference order")
        assert np.mean(g.order == r["order"]) > 0.999, (context, "output order vs the reference order")
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
    # constant ims column (the normal case without ion mobility): the elimination meets exact zeros and rounding noise among
    # its pivot candidates; the device must take the oracle's branch whatever it is, on every one of these data sets
    tol = Tolerance("ppm", -10.0, 10.0)
    branches = set()
    for n, seed in ((4000, 3), (900, 14), (12000, 15), (50000, 16), (2500, 17)):
        f, pk, npk, prk, npr = synthetic_features(n, seed=seed, zero_ims=True)
        g, _ = compare(f, tol, pk, npk, prk, npr, f"zero ims n={n} seed={seed}")
        branches.add(bool(g.lda_fitted))
    assert branches  # (which branches occur is data; both are legitimate)


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
    g, o = compare(flat, tol, pk, npk, prk, npr, "end to end")  # (asserts lda_fitted equal: ims is constant here)
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
