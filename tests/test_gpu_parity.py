"""GPU parity tests: the HIP path (through the C ABI) against the oracle on identical inputs."""
import numpy as np
import pytest

import oracle_lib
from parity_utils import assert_features_equal, assert_initial_hits_equal
from sage_amd import _lib as L
from sage_amd.api import (DatabaseParameters, DeviceDatabase, RawBatch, RawSpectrum, Scorer, ScorerParams, SpectrumBatch,
                          SpectrumProcessor, Tolerance)
from sage_amd.synthetic import synthetic_fasta, synthetic_spectra
from test_oracle_golden import c1_batch, integration_scorer

pytestmark = pytest.mark.gpu


def product_process(top_n, deiso, min_mz, mz, it, z):
    out = SpectrumProcessor(top_n, deiso, min_mz).process(RawSpectrum(mz, it, 0.0, z or None))
    return out.masses, out.intensities, out.total_ion_current


class World:
    def __init__(self, fasta, db_params, spectra_kwargs, n_spectra, seed, max_peaks=150):
        self.host = db_params.build(fasta)
        self.dev = DeviceDatabase(self.host, 0)
        self.orc = oracle_lib.OracleDb.from_product(self.host)
        raw = synthetic_spectra(self.host, n_spectra, seed, **spectra_kwargs)
        sp = SpectrumProcessor(max_peaks, True, 0.0)
        self.batch = SpectrumBatch.from_spectra([sp.process(r) for r in raw])

    def check(self, params, context, hits=True, batch=None, every=1, dev=None):
        # OpenMSHyperScore goes through f32 ln_1p: device libm and glibc may differ by an f32 ulp (~6e-8 relative).
        # SageHyperScore uses f64 ln only — correctly rounded on both sides: the f64 fields are held to EQUALITY
        # (parity_utils.assert_features_equal).  North-star tolerance: 1e-4.
        rel_tol = 1e-6 if params.score_type == "OpenMSHyperScore" else None
        batch = batch or self.batch
        scorer = Scorer(dev or self.dev, params)
        dbatch = scorer.upload(batch)
        if hits:
            assert_initial_hits_equal(scorer, dbatch, self.orc, params, batch, context, every)
        gf, gc = scorer.score_resident(dbatch)
        gf, gc = gf.copy(), gc.copy()
        t_first = scorer.last_timing()
        of, oc, _, _ = self.orc.score(params, batch)
        n = assert_features_equal(gf, gc, of, oc, context, rel_tol)
        # once more on the same handle: a narrow-search step launches its exact retry pass only when the step before had something
        # to retry (the first step of a handle finds out from the counters and launches it late) — both routes, same records
        gfb, gcb = scorer.score_resident(dbatch)
        assert np.array_equal(gc, gcb) and gf[np.arange(gf.shape[1])[None, :] < gc[:, None]].tobytes() == \
            gfb[np.arange(gf.shape[1])[None, :] < gc[:, None]].tobytes(), f"{context}: the second step on the same scorer differs"
        assert scorer.last_timing()["n_retry"] == t_first["n_retry"], context
        gf2, gc2 = scorer.score(batch)  # the upload+score+download entry point
        assert_features_equal(gf2, gc2, of, oc, context + " (score_batch)", rel_tol)
        return n, t_first


@pytest.fixture(scope="module")
def small_world(gpu_required):
    fasta = synthetic_fasta(300, seed=11)
    params = DatabaseParameters(bucket_size=2048, enzyme=dict(missed_cleavages=1, cleave_at="KR", restrict="P"),
                                static_mods={"C": 57.0215}, variable_mods={"M": [15.9949]})
    return World(fasta, params, {}, 600, seed=21)


def test_c1_known_answer_on_gpu(gpu_required):
    """crates/sage-cli/tests/integration.rs:7-52 through the HIP path: 1 PSM, matched_peaks == 21."""
    d, batch = c1_batch(product_process)
    host = DatabaseParameters().build(d["fasta"])
    dev = DeviceDatabase(host, 0)
    params = integration_scorer()
    scorer = Scorer(dev, params)
    feats, counts = scorer.score(batch)
    assert counts[0] == 1 and feats[0, 0]["matched_peaks"] == 21
    assert host.peptide_string(int(feats[0, 0]["peptide_idx"])) == "LQSRPAAPPAPGPGQLTLR"
    orc = oracle_lib.OracleDb.from_product(host)
    of, oc, _, _ = orc.score(params, batch)
    assert_features_equal(feats, counts, of, oc, "C1")
    assert_initial_hits_equal(scorer, scorer.upload(batch), orc, params, batch, "C1")


def test_hyperscores_against_the_platform_libm(small_world):
    """The product's ln is correctly rounded (crlog.h); every other parity test compares with the oracle in its correctly rounded
    mode, bit for bit.  Here the oracle calls the platform libm instead — what the reference itself does on this host (glibc 2.35:
    0.52 ulp): the f64 fields agree within an ulp or two and the overwhelming majority are equal (parity_utils.MIN_EQUAL)."""
    w = small_world
    params = ScorerParams(report_psms=5, min_matched_peaks=1, precursor_tol=Tolerance("da", -3.0, 3.0))
    scorer = Scorer(w.dev, params)
    gf, gc = scorer.score_resident(scorer.upload(w.batch))
    with oracle_lib.LogMode(0):
        of, oc, _, _ = w.orc.score(params, w.batch)
        n = assert_features_equal(gf, gc, of, oc, "platform libm")
    assert n > 1000
    of1, oc1, _, _ = w.orc.score(params, w.batch)
    assert_features_equal(gf, gc, of1, oc1, "correctly rounded")


def test_narrow_search_known_charge(small_world):
    n, t = small_world.check(ScorerParams(), "narrow ±10ppm")
    assert n > 300  # most non-noise spectra are identified
    assert t["n_wide"] == 0


@pytest.mark.parametrize("variant", ["stream", "probe"])
def test_narrow_kernel_variants(small_world, monkeypatch, variant):
    """Both fragment-matching strategies of the narrow kernel (peptide-major stream / per-peak table lookups; the C ABI
    normally picks one per batch from the mean window size), with windows from a handful to ~1000 candidates,
    isotope folding and unknown charge."""
    monkeypatch.setenv("SAGE_HIP_NARROW", variant)
    small_world.check(ScorerParams(), f"{variant}: narrow ±10ppm")
    small_world.check(ScorerParams(precursor_tol=Tolerance("da", -1.5, 1.5), report_psms=3), f"{variant}: ±1.5 Da")
    small_world.check(ScorerParams(min_isotope_err=-1, max_isotope_err=3, precursor_tol=Tolerance("ppm", -20.0, 20.0),
                                   max_fragment_charge=3), f"{variant}: isotope -1..3, fragment charge 3")
    b = small_world.batch
    unknown = SpectrumBatch(b.peak_off, b.masses, b.intensities, b.precursor_mz, np.zeros(b.n, np.uint8), b.total_ion_current)
    small_world.check(ScorerParams(precursor_tol=Tolerance("da", -0.5, 0.5)), f"{variant}: charge None", batch=unknown)


@pytest.mark.parametrize("no_sched", [None, "1"])
def test_schedule_records_and_the_order_array(small_world, monkeypatch, no_sched):
    """prelim_kernel / rescore_kernel start a block from the batch's schedule records (DevBatchView::sched: spectrum, peak range,
    charge, m/z, isolation window in schedule order — built at upload) or, with SAGE_HIP_NO_SCHED=1 and in every retry pass, from
    order[b] and the per-spectrum arrays: the same Features either way — with isolation windows (wide-window search), unknown
    charges, isotope folding, and in parts on several streams."""
    if no_sched:
        monkeypatch.setenv("SAGE_HIP_NO_SCHED", no_sched)
    small_world.check(ScorerParams(report_psms=2), f"no_sched={no_sched}: narrow")
    small_world.check(ScorerParams(min_isotope_err=-1, max_isotope_err=2, precursor_tol=Tolerance("ppm", -20.0, 20.0)),
                      f"no_sched={no_sched}: isotope -1..2")
    b = small_world.batch
    unknown = SpectrumBatch(b.peak_off, b.masses, b.intensities, b.precursor_mz, np.zeros(b.n, np.uint8), b.total_ion_current)
    small_world.check(ScorerParams(precursor_tol=Tolerance("da", -0.5, 0.5)), f"no_sched={no_sched}: charge None", batch=unknown)
    rng = np.random.default_rng(11)
    lo = -rng.uniform(0.3, 1.2, b.n).astype(np.float32)
    hi = rng.uniform(0.3, 1.2, b.n).astype(np.float32)
    lo[::7] = np.nan  # (spectrum.rs: no isolation window recorded -> the +-2.4 default of scoring.rs:430)
    windows = SpectrumBatch(b.peak_off, b.masses, b.intensities, b.precursor_mz, b.precursor_charge, b.total_ion_current, lo, hi)
    small_world.check(ScorerParams(wide_window=True, chimera=True, report_psms=2), f"no_sched={no_sched}: wide window", batch=windows)
    monkeypatch.setenv("SAGE_HIP_WAYS", "3")
    small_world.check(ScorerParams(), f"no_sched={no_sched}: three parts")


def test_rescoring_with_and_without_cooperative_matching(small_world, monkeypatch):
    """score_candidate takes a candidate with many filter hits with the whole wavefront (lookups in parallel, sums in item
    order) when few candidates are that heavy; SAGE_HIP_DEBUG_FLAGS=32 leaves every candidate to its own lane, 64 takes every
    heavy candidate together.  Same Features either way, also with fragment
    charges above 3 (unfiltered items) and a wide fragment tolerance (every bin of the peak bitmap set)."""
    for flags in (None, "32", "64"):
        if flags:
            monkeypatch.setenv("SAGE_HIP_DEBUG_FLAGS", flags)
        small_world.check(ScorerParams(report_psms=3), f"coop flags={flags}: narrow")
        small_world.check(ScorerParams(max_fragment_charge=5, override_precursor_charge=True, min_precursor_charge=5,
                                       max_precursor_charge=6, precursor_tol=Tolerance("da", -3.0, 3.0)),
                          f"coop flags={flags}: fragment charges 1..5")
        small_world.check(ScorerParams(fragment_tol=Tolerance("da", -0.6, 0.6), chimera=True, report_psms=2),
                          f"coop flags={flags}: ±0.6 Da fragments, chimera")


def test_rescoring_with_the_dense_work_list_and_the_walk(small_world, monkeypatch):
    """The lanes' own filter hits of a scoring round's last 64-ion chunk go through select_most_intense_peak 64 items at a time
    (a work list in the bitmap's bytes: kernels.hip, score_candidates) unless another round will read the bitmap again (chimera),
    another chunk follows, a fragment charge above 3 is unfiltered or the list overflows; SAGE_HIP_DEBUG_FLAGS=128 takes the
    lane-by-lane walk everywhere, 256 caps the list at 64 items (spectra on both routes), and with 32 the heavy candidates'
    dozens of hits go through the list as well.  Same Features on every route."""
    for flags in (None, "128", "256", "32", "288"):
        if flags:
            monkeypatch.setenv("SAGE_HIP_DEBUG_FLAGS", flags)
        small_world.check(ScorerParams(report_psms=3), f"dense flags={flags}: narrow")
        small_world.check(ScorerParams(max_fragment_charge=3, precursor_tol=Tolerance("da", -3.0, 3.0), report_psms=2,
                                       override_precursor_charge=True, min_precursor_charge=3, max_precursor_charge=4),
                          f"dense flags={flags}: fragment charges 1..3, ±3 Da")
        small_world.check(ScorerParams(fragment_tol=Tolerance("da", -0.6, 0.6), min_matched_peaks=2),
                          f"dense flags={flags}: ±0.6 Da fragments (every bin of the bitmap set: lists beyond the cap)")
        small_world.check(ScorerParams(max_fragment_charge=5, override_precursor_charge=True, min_precursor_charge=5,
                                       max_precursor_charge=6), f"dense flags={flags}: fragment charges above 3 (the walk)")


def test_rescoring_with_short_and_ieee_divisions(small_world, monkeypatch):
    """Tolerance::bounds (mass.rs:21-35) and the m/z of charge-3 fragments (scoring.rs:707) in the rescoring kernel: the instance
    with the short division forms (core.h: div_const_fast; opt-in, SAGE_HIP_SHORT_DIVISIONS=1) where the host has bounded the
    dividends (capi.hip: scorer_tol_mode — ppm tolerances with non-zero bounds over a sane ion table), the IEEE sequence otherwise
    (a zero bound) and by default.  Same Features either way — the oracle's."""
    for short in (None, "1"):
        if short:
            monkeypatch.setenv("SAGE_HIP_SHORT_DIVISIONS", short)
        small_world.check(ScorerParams(report_psms=1), f"short={short}: ppm[-10, 10]")
        small_world.check(ScorerParams(fragment_tol=Tolerance("ppm", -20.0, 5.0)), f"short={short}: ppm[-20, 5]")
        small_world.check(ScorerParams(fragment_tol=Tolerance("ppm", 0.0, 20.0)), f"short={short}: ppm[0, 20] (a zero bound)")
        small_world.check(ScorerParams(fragment_tol=Tolerance("pct", -0.002, 0.002)), f"short={short}: pct")
        small_world.check(ScorerParams(fragment_tol=Tolerance("da", -0.02, 0.02), max_fragment_charge=3, override_precursor_charge=True,
                                       min_precursor_charge=4, max_precursor_charge=4, precursor_tol=Tolerance("da", -2.0, 2.0)),
                          f"short={short}: Da fragments, fragment charges 1..3")
        small_world.check(ScorerParams(max_fragment_charge=4, override_precursor_charge=True, min_precursor_charge=4,
                                       max_precursor_charge=5, precursor_tol=Tolerance("da", -3.0, 3.0),
                                       fragment_tol=Tolerance("ppm", -15.0, 15.0)), f"short={short}: fragment charges 1..4")
        small_world.check(ScorerParams(chimera=True, report_psms=3, fragment_tol=Tolerance("ppm", -7.0, 12.0), max_fragment_charge=3),
                          f"short={short}: chimera, ppm[-7, 12]")


def test_report_psms_and_score_types(small_world):
    small_world.check(ScorerParams(report_psms=5, precursor_tol=Tolerance("ppm", -50.0, 50.0)), "report_psms=5")
    small_world.check(ScorerParams(score_type="OpenMSHyperScore", min_matched_peaks=2), "OpenMS score")
    small_world.check(ScorerParams(report_psms=30, precursor_tol=Tolerance("da", -3.0, 3.0)), "report_psms=30 (k=60)")


def test_report_psms_beyond_a_wavefront(small_world, monkeypatch):
    """report_psms > 32: trim_hits keeps k = max(50, 2 * report_psms) > 64 candidates (scoring.rs:322-329), preliminary lists wider
    than a wavefront — the BIGK kernels (heaps in LDS, every trim exact, rescoring 64 candidates at a time).  Lists (initial_hits)
    and PSMs equal to the oracle's for 50, 100 and 128 PSMs per spectrum: narrow and large windows, mixed routing, isotope errors x
    charges (folded lists), chimera, wide windows, quick_score; resident and streamed."""
    w = small_world
    b = w.batch
    wide = Tolerance("da", -20.0, 20.0)  # windows of a few hundred candidates: longer than k, inside the narrow kernel's counters
    n, t = w.check(ScorerParams(report_psms=50, precursor_tol=wide), "report_psms=50 (k=100), +-20 Da")
    assert t["n_retry"] == 0
    # (min_matched_peaks = 1: most of a list passes scoring.rs:491, so the ranks beyond 32 and 64 are really reported)
    n, t = w.check(ScorerParams(report_psms=50, precursor_tol=wide, min_matched_peaks=1), "report_psms=50, min_matched_peaks=1")
    assert n > b.n * 35
    n, t = w.check(ScorerParams(report_psms=100, precursor_tol=wide, min_matched_peaks=1), "report_psms=100 (k=200), +-20 Da")
    assert n > b.n * 40
    # (min_matched_peaks = 1: a spectrum of this case holds candidates that match the same single peak as a b- and as a y-ion —
    # hyperscores ln(i) + lnfact(1) + lnfact(0) and ln(i) + lnfact(0) + lnfact(1), scoring.rs:163-201, two DIFFERENT sums that land
    # on one double or on two neighbours depending on the last bit of ln(i).  Round 3's device ln (ocml) differed from the host's
    # there and ordered such pairs differently among the 128 reported PSMs; with the correctly rounded ln on both sides
    # (crlog.h / the oracle's libquadmath mode) hyperscores and order are equal.)
    n, t = w.check(ScorerParams(report_psms=128, precursor_tol=Tolerance("da", -60.0, 60.0), min_matched_peaks=1,
                                fragment_tol=Tolerance("da", -0.3, 0.3)),
                   "report_psms=128 (k=256), +-60 Da, min_matched_peaks=1", batch=b.subset(np.arange(0, b.n, 2)))
    assert n > (b.n // 2) * 64  # (on average more PSMs per spectrum than a wavefront has lanes)
    w.check(ScorerParams(report_psms=40), "report_psms=40, +-10 ppm (windows shorter than k)")
    unknown = SpectrumBatch(b.peak_off, b.masses, b.intensities, b.precursor_mz, np.zeros(b.n, np.uint8), b.total_ion_current,
                            b.isolation_lo, b.isolation_hi, b.scan_start_time, b.inverse_ion_mobility, b.file_id)
    w.check(ScorerParams(report_psms=50, precursor_tol=Tolerance("da", -8.0, 8.0), min_isotope_err=-1, max_isotope_err=2),
            "report_psms=50, iso -1..2 x charges 2..4", batch=unknown.subset(np.arange(0, b.n, 2)))
    # large windows: the tile kernels' seeds / heaps of k entries, replayed in LDS; and both routes in one batch
    n, t = w.check(ScorerParams(report_psms=50, precursor_tol=Tolerance("da", -300.0, 300.0)), "report_psms=50, open search",
                   batch=b.subset(np.arange(0, b.n, 4)))
    assert t["n_wide"] > 100
    w.check(ScorerParams(report_psms=100, precursor_tol=Tolerance("da", -150.0, 150.0), min_isotope_err=0, max_isotope_err=1),
            "report_psms=100, +-150 Da x iso 0..1", batch=unknown.subset(np.arange(0, b.n, 6)))
    monkeypatch.setenv("SAGE_HIP_WCAP", "128")
    n, t = w.check(ScorerParams(report_psms=50, precursor_tol=wide), "report_psms=50, mixed routing")
    assert 0 < t["n_wide"] < b.n
    monkeypatch.delenv("SAGE_HIP_WCAP")
    w.check(ScorerParams(report_psms=40, chimera=True, precursor_tol=wide, min_matched_peaks=2), "report_psms=40, chimera",
            batch=b.subset(np.arange(0, b.n, 3)))
    for low_memory in (False, True):
        params = ScorerParams(report_psms=50, precursor_tol=wide)
        scorer = Scorer(w.dev, params)
        gk = scorer.quick_score(scorer.upload(b), low_memory)
        np.testing.assert_array_equal(gk, w.orc.quick_score(params, b, low_memory), err_msg=f"quick_score report_psms=50 low_memory={low_memory}")
    # the other score type, a capped fragment charge, and the Fragments of 40 PSMs per spectrum
    w.check(ScorerParams(report_psms=50, precursor_tol=wide, score_type="OpenMSHyperScore", min_matched_peaks=2, max_fragment_charge=1),
            "report_psms=50, OpenMS score, fragment charge 1", batch=b.subset(np.arange(0, b.n, 3)))
    _check_annotation(w, ScorerParams(annotate_matches=True, report_psms=40, precursor_tol=wide, min_matched_peaks=2),
                      b.subset(np.arange(0, b.n, 4)), "annotate, report 40")
    # report_psms above 128 (round 4: lists, heaps and per-candidate scores of up to 1024 entries — a compute unit's whole LDS for
    # the instances that keep them there): 500 PSMs per spectrum, narrow windows of a few thousand candidates (WCAP raised so that
    # they stay in the narrow kernel), large windows through the tile kernels, both routes in one batch, chimera rounds
    few = b.subset(np.arange(0, b.n, 6))
    monkeypatch.setenv("SAGE_HIP_WCAP", "8192")
    n, t = w.check(ScorerParams(report_psms=500, precursor_tol=Tolerance("da", -150.0, 150.0), min_matched_peaks=1,
                                fragment_tol=Tolerance("da", -0.3, 0.3)), "report_psms=500 (k=1000), narrow kernel", batch=few)
    assert n > few.n * 150 and t["n_wide"] < few.n // 2  # (most windows in the narrow kernel, the widest in the tile kernels)
    monkeypatch.delenv("SAGE_HIP_WCAP")
    n, t = w.check(ScorerParams(report_psms=500, precursor_tol=Tolerance("da", -150.0, 150.0), min_matched_peaks=1,
                                fragment_tol=Tolerance("da", -0.3, 0.3)), "report_psms=500, tile kernels", batch=few)
    assert n > few.n * 150 and t["n_wide"] > few.n // 2
    w.check(ScorerParams(report_psms=300, precursor_tol=Tolerance("da", -40.0, 40.0), min_matched_peaks=1), "report_psms=300, mixed sizes",
            batch=few)
    w.check(ScorerParams(report_psms=200, chimera=True, precursor_tol=wide, min_matched_peaks=3), "report_psms=200, chimera",
            batch=b.subset(np.arange(0, b.n, 40)))
    for low_memory in (True, False):
        params = ScorerParams(report_psms=400, precursor_tol=Tolerance("da", -60.0, 60.0))
        scorer = Scorer(w.dev, params)
        gk = scorer.quick_score(scorer.upload(few), low_memory)
        np.testing.assert_array_equal(gk, w.orc.quick_score(params, few, low_memory), err_msg=f"quick_score report_psms=400 low_memory={low_memory}")
    # folded lists of 15 queries per spectrum x 1000 candidates (110 KB of LDS) still fit a compute unit ...
    unknown_few = unknown.subset(np.arange(0, b.n, 50))
    w.check(ScorerParams(report_psms=500, precursor_tol=Tolerance("da", -30.0, 30.0), min_isotope_err=-1, max_isotope_err=3, min_matched_peaks=1),
            "report_psms=500, iso -1..3 x charges 2..4", batch=unknown_few)
    # ... and where they do not (round 5; rounds 3-4 refused): the lists, heaps and per-candidate arrays of the same kernels in a
    # global-memory workspace, a slice per workgroup of the capped grids — eleven isotope errors x 500 PSMs; 1100 PSMs per
    # spectrum, twice the old cap of 512 (scoring.rs:322-329 has none), narrow, tiled, folded, chimera, quick_score
    w.check(ScorerParams(report_psms=500, precursor_tol=Tolerance("da", -30.0, 30.0), min_isotope_err=-1, max_isotope_err=9, min_matched_peaks=1),
            "report_psms=500, iso -1..9 x charges 2..4: lists in global memory", batch=unknown_few)
    monkeypatch.setenv("SAGE_HIP_WCAP", "8192")
    n, t = w.check(ScorerParams(report_psms=1100, precursor_tol=Tolerance("da", -300.0, 300.0), min_matched_peaks=1,
                                fragment_tol=Tolerance("da", -0.3, 0.3)), "report_psms=1100 (k=2200), narrow kernel", batch=few)
    assert n > few.n * 300
    monkeypatch.delenv("SAGE_HIP_WCAP")
    n, t = w.check(ScorerParams(report_psms=1100, precursor_tol=Tolerance("da", -300.0, 300.0), min_matched_peaks=1,
                                fragment_tol=Tolerance("da", -0.3, 0.3)), "report_psms=1100, tile kernels", batch=few)
    assert n > few.n * 300 and t["n_wide"] > few.n // 2
    w.check(ScorerParams(report_psms=600, precursor_tol=Tolerance("da", -100.0, 100.0), min_isotope_err=0, max_isotope_err=1, min_matched_peaks=1),
            "report_psms=600, iso 0..1 x charges 2..4, folded", batch=unknown_few)
    w.check(ScorerParams(report_psms=520, chimera=True, precursor_tol=Tolerance("da", -200.0, 200.0), min_matched_peaks=3), "report_psms=520, chimera",
            batch=b.subset(np.arange(0, b.n, 100)))
    for low_memory in (True, False):
        params = ScorerParams(report_psms=700, precursor_tol=Tolerance("da", -200.0, 200.0))
        scorer = Scorer(w.dev, params)
        gk = scorer.quick_score(scorer.upload(few), low_memory)
        np.testing.assert_array_equal(gk, w.orc.quick_score(params, few, low_memory), err_msg=f"quick_score report_psms=700 low_memory={low_memory}")
    # the workspace forced on lists that would fit LDS (every wide-list test above through the other memory)
    monkeypatch.setenv("SAGE_HIP_FORCE_HUGE", "1")
    w.check(ScorerParams(report_psms=50, precursor_tol=Tolerance("da", -8.0, 8.0), min_isotope_err=-1, max_isotope_err=2),
            "report_psms=50, folded, forced workspace", batch=unknown.subset(np.arange(0, b.n, 2)))
    w.check(ScorerParams(report_psms=100, precursor_tol=Tolerance("da", -150.0, 150.0)), "report_psms=100, tiles, forced workspace", batch=few)
    monkeypatch.delenv("SAGE_HIP_FORCE_HUGE")
    with pytest.raises(L.SageHipError):
        Scorer(w.dev, ScorerParams(report_psms=40000))


def test_isotope_errors_and_fragment_charge(small_world):
    small_world.check(ScorerParams(min_isotope_err=-1, max_isotope_err=3, precursor_tol=Tolerance("ppm", -20.0, 20.0)),
                      "isotope -1..3")
    small_world.check(ScorerParams(min_isotope_err=1, max_isotope_err=1), "isotope 1..1 quirk (searches 0)")
    small_world.check(ScorerParams(max_fragment_charge=1), "max_fragment_charge=1")
    small_world.check(ScorerParams(max_fragment_charge=3, fragment_tol=Tolerance("da", -0.02, 0.02)), "frag Da tol")
    small_world.check(ScorerParams(fragment_tol=Tolerance("pct", -0.001, 0.001)), "frag pct tol")


@pytest.mark.parametrize("variant", ["stream", "probe"])
def test_asymmetric_and_one_sided_tolerances(small_world, monkeypatch, variant):
    """Tolerance::bounds takes any (lo, hi) (mass.rs:21-35): windows that lean to one side of the centre or lie entirely beside
    it — on the fragment side (the position-table cells of both matching variants, the peak bitmap's reach max(|lo|, |hi|), the
    general branch of the symmetric-ppm shortcut) and on the precursor side (the window query through pep_lut)."""
    monkeypatch.setenv("SAGE_HIP_NARROW", variant)
    w = small_world
    for ftol in (Tolerance("ppm", -20.0, 5.0), Tolerance("ppm", 5.0, 20.0), Tolerance("da", 0.01, 0.3), Tolerance("da", -0.3, -0.01),
                 Tolerance("pct", -0.002, 0.0005)):
        n, t = w.check(ScorerParams(fragment_tol=ftol, min_matched_peaks=2), f"{variant}: fragment_tol {ftol}")
        assert n > (200 if ftol.lo < 0.0 < ftol.hi else 0), (str(ftol), n)  # (beside the centre: chance matches only, 3 ppm noise)
    n, t = w.check(ScorerParams(precursor_tol=Tolerance("ppm", -50.0, 10.0)), f"{variant}: precursor_tol ppm[-50, 10]")
    assert n > 300
    w.check(ScorerParams(precursor_tol=Tolerance("ppm", 5.0, 60.0), fragment_tol=Tolerance("ppm", -20.0, 5.0), report_psms=3),
            f"{variant}: both one-sided / asymmetric, three PSMs")
    w.check(ScorerParams(precursor_tol=Tolerance("da", -4.0, 0.5), fragment_tol=Tolerance("ppm", 5.0, 20.0), chimera=True, report_psms=2,
                         max_fragment_charge=3), f"{variant}: da[-4, 0.5], chimera")
    # large windows (the tiled kernels' own table lookups) with an asymmetric fragment tolerance
    sub = w.batch.subset(np.arange(0, w.batch.n, 6))
    n, t = w.check(ScorerParams(precursor_tol=Tolerance("da", -300.0, 100.0), fragment_tol=Tolerance("ppm", -20.0, 5.0)),
                   f"{variant}: open da[-300, 100], fragments ppm[-20, 5]", batch=sub)
    assert t["n_wide"] > 0


def test_unknown_and_overridden_precursor_charge(small_world):
    b = small_world.batch
    unknown = SpectrumBatch(b.peak_off, b.masses, b.intensities, b.precursor_mz, np.zeros(b.n, np.uint8),
                            b.total_ion_current, b.isolation_lo, b.isolation_hi, b.scan_start_time,
                            b.inverse_ion_mobility, b.file_id)
    small_world.check(ScorerParams(), "charge None -> 2..4", batch=unknown)
    small_world.check(ScorerParams(override_precursor_charge=True, min_precursor_charge=1, max_precursor_charge=5,
                                   min_isotope_err=-1, max_isotope_err=1), "override charge 1..5 x iso -1..1")


@pytest.mark.parametrize("queue_later", ["0", "1"])
def test_wide_tolerance_hits_large_window_path(small_world, monkeypatch, queue_later):
    """±150 Da on a small database: the window exceeds the LDS counter capacity for many spectra, so the
    large-window kernel runs; results must still be identical — with the spectra queued by prelim_kernel itself (one atomic each)
    and by queue_kernel behind it (round 6; the default depends on the batch's widest window)."""
    monkeypatch.setenv("SAGE_HIP_QUEUE_LATER", queue_later)
    idx = np.arange(0, small_world.batch.n, 6)
    sub = small_world.batch.subset(idx)
    n, t = small_world.check(ScorerParams(precursor_tol=Tolerance("da", -300.0, 300.0)), "open ±300 Da", batch=sub)
    assert t["n_wide"] > 0
    # a mixed batch: narrow spectra beside queued ones
    monkeypatch.setenv("SAGE_HIP_WCAP", "64")
    n, t = small_world.check(ScorerParams(precursor_tol=Tolerance("da", -2.0, 2.0)), "mixed narrow / queued", batch=sub)
    assert 0 < t["n_wide"] < sub.n


@pytest.mark.parametrize("tile_shift,replay", [(11, "wave"), (12, "lane"), (15, "wave"), (15, "lane"), (12, "both")])
def test_large_window_tile_kernel(small_world, monkeypatch, tile_shift, replay):
    """The tiled large-window kernel with small tiles (2048 / 4096 peptides) so that every window spans many
    tiles, the first k slots straddle tile boundaries, and partial first/last tiles occur: ±500 Da, isotope
    folding, unknown charge, report_psms > 1 — every branch of the nested k-selects."""
    monkeypatch.setenv("SAGE_HIP_TILE_SHIFT", str(tile_shift))
    monkeypatch.setenv("SAGE_HIP_WCAP", "64")
    if replay == "lane":  # the heap replay kernel with one lane per query (since round 6 only where these knobs ask for it)
        monkeypatch.setenv("SAGE_HIP_REPLAY_WAVE_MAX", "0")
    if replay == "both":  # ... and the two side by side: streams above 64 words by wavefront, the rest a lane each
        monkeypatch.setenv("SAGE_HIP_REPLAY_WAVE_MAX", "4")
        monkeypatch.setenv("SAGE_HIP_REPLAY_LANE_MAX", "64")
        # (and the count kernel's run-start marks cover one round of cells only: every further round of a unit re-marks)
        monkeypatch.setenv("SAGE_HIP_DEBUG_FLAGS", "2048")
    dev = DeviceDatabase(small_world.host, 0)
    idx = np.arange(0, small_world.batch.n, 5)
    sub = small_world.batch.subset(idx)
    n, t = small_world.check(ScorerParams(precursor_tol=Tolerance("da", -500.0, 100.0)), "open -500/+100 Da", batch=sub, dev=dev)
    assert t["n_wide"] > 0 and n > 50
    small_world.check(ScorerParams(precursor_tol=Tolerance("da", -40.0, 40.0), min_isotope_err=-1, max_isotope_err=2,
                                   report_psms=4), "open ±40 Da x iso -1..2", batch=sub, dev=dev)
    b = sub
    unknown = SpectrumBatch(b.peak_off, b.masses, b.intensities, b.precursor_mz, np.zeros(b.n, np.uint8),
                            b.total_ion_current)
    small_world.check(ScorerParams(precursor_tol=Tolerance("da", -60.0, 60.0), max_fragment_charge=2), "open, charge None",
                      batch=unknown, dev=dev)
    # windows around the narrow kernel's capacity: both kernels in one batch
    n, t = small_world.check(ScorerParams(precursor_tol=Tolerance("da", -2.0, 2.0)), "mixed narrow / tiled routing",
                             batch=sub, dev=dev)
    assert 0 < t["n_wide"] < sub.n
    # half-Dalton fragment windows: runs of dozens of cells, units of several thousand — more than a thread's cells in flight (the
    # synchronous rounds of a unit) and, with SAGE_HIP_DEBUG_FLAGS=2048 (the "both" case), more than the run-start marks cover at a
    # time (the count kernel's re-mark path; a build with that path left empty fails here: scripts/experiments/r06_lab/gpu_r7u.sh)
    small_world.check(ScorerParams(precursor_tol=Tolerance("da", -300.0, 300.0), fragment_tol=Tolerance("da", -0.5, 0.5), report_psms=2),
                      "open ±300 Da, fragment ±0.5 Da", batch=sub.subset(np.arange(0, sub.n, 3)), dev=dev)
    small_world.check(ScorerParams(precursor_tol=Tolerance("da", -5000.0, 5000.0), min_matched_peaks=1,
                                   report_psms=30), "every peptide in the window, k = 60", batch=small_world.batch.subset(np.arange(0, 40)),
                      dev=dev)
    # the replay kernel's 64-bit key path (taken by itself for windows above 2^21 slots or counts >= 63)
    monkeypatch.setenv("SAGE_HIP_DEBUG_FLAGS", "2")
    small_world.check(ScorerParams(precursor_tol=Tolerance("da", -500.0, 100.0), report_psms=3), "open, 64-bit replay keys",
                      batch=sub, dev=dev)
    # the first pass counts in u8 and hands spectra whose counters might wrap to the u16 retry pass: force that hand-over for
    # every slot with three matches, and switch the u8 instance off altogether
    monkeypatch.setenv("SAGE_HIP_DEBUG_FLAGS", "16")
    n, t = small_world.check(ScorerParams(precursor_tol=Tolerance("da", -500.0, 100.0), report_psms=3), "open, u8 counter overflow path",
                             batch=sub, dev=dev)
    assert t["n_retry"] > sub.n // 2
    # (the retry pass above counted the flagged spectra again, into the first pass's slots; now without any reuse of the first
    # pass's counts: the retry pass counts everything itself)
    monkeypatch.setenv("SAGE_HIP_NO_REUSE", "1")
    small_world.check(ScorerParams(precursor_tol=Tolerance("da", -500.0, 100.0), report_psms=3), "open, overflow path, no reuse",
                      batch=sub, dev=dev)
    monkeypatch.delenv("SAGE_HIP_DEBUG_FLAGS")
    small_world.check(ScorerParams(precursor_tol=Tolerance("da", -40.0, 40.0), min_isotope_err=-1, max_isotope_err=2, report_psms=4),
                      "open ±40 Da x iso -1..2, no reuse", batch=sub, dev=dev)
    monkeypatch.delenv("SAGE_HIP_NO_REUSE")
    monkeypatch.setenv("SAGE_HIP_NO_U8", "1")
    n, t2 = small_world.check(ScorerParams(precursor_tol=Tolerance("da", -500.0, 100.0), report_psms=3), "open, u16 counters only",
                              batch=sub, dev=dev)
    assert t2["n_retry"] < t["n_retry"]
    if tile_shift == 12:
        # more precursor-window queries per spectrum than the count kernel searches up front (kernels.hip: TILE_QW_MAX = 64): 21 isotope
        # errors x 4 charge states = 84 — the windows are then searched query by query, as before round 6
        monkeypatch.delenv("SAGE_HIP_NO_U8")
        small_world.check(ScorerParams(precursor_tol=Tolerance("da", -30.0, 30.0), min_isotope_err=-10, max_isotope_err=10,
                                       min_precursor_charge=1, max_precursor_charge=4, report_psms=2), "84 queries per spectrum",
                          batch=SpectrumBatch(unknown.peak_off, unknown.masses, unknown.intensities, unknown.precursor_mz,
                                              unknown.precursor_charge, unknown.total_ion_current).subset(np.arange(0, 24)), dev=dev)
    dev.close()


def test_chimera_and_wide_window(gpu_required):
    fasta = synthetic_fasta(150, seed=12)
    params = DatabaseParameters(bucket_size=1024, enzyme=dict(missed_cleavages=1, cleave_at="KR", restrict="P"),
                                static_mods={"C": 57.0215})
    w = World(fasta, params, dict(chimeric=3, isolation_half_width=6.0, annotate_charge=False), 200, seed=22)
    w.check(ScorerParams(chimera=True, report_psms=5), "chimera, charge None")
    w.check(ScorerParams(chimera=True, wide_window=True, report_psms=5), "chimera + wide_window (DIA-style)")
    w.check(ScorerParams(wide_window=True, report_psms=3), "wide_window only")
    w.check(ScorerParams(chimera=True, wide_window=True, report_psms=40), "chimera + wide_window, 40 rounds (lists of 80 candidates)")
    w.check(ScorerParams(wide_window=True, report_psms=64), "wide_window, report_psms=64 (k=128)")
    no_iso = SpectrumBatch(w.batch.peak_off, w.batch.masses, w.batch.intensities, w.batch.precursor_mz,
                           w.batch.precursor_charge, w.batch.total_ion_current)
    w.check(ScorerParams(wide_window=True), "wide_window without isolation window (Da ±2.4 default)", batch=no_iso)


def test_edge_cases(small_world):
    b = small_world.batch
    host = small_world.host
    # empty spectrum, one-peak spectrum, precursor below / above every peptide, duplicated peaks
    masses = [np.zeros(0, np.float32), np.array([500.0], np.float32), b.masses[:100].copy(), b.masses[:100].copy(),
              np.repeat(b.masses[int(b.peak_off[5]):int(b.peak_off[6])], 2)]
    intens = [np.zeros(0, np.float32), np.array([10.0], np.float32), b.intensities[:100].copy(),
              b.intensities[:100].copy(), np.repeat(b.intensities[int(b.peak_off[5]):int(b.peak_off[6])], 2)]
    masses[2] = np.sort(masses[2]); masses[3] = np.sort(masses[3])
    prec = [600.0, 600.0, 50.0, 9000.0, float(b.precursor_mz[5])]
    z = [2, 2, 2, 2, int(b.precursor_charge[5])]
    off = np.cumsum([0] + [len(m) for m in masses])
    eb = SpectrumBatch(off, np.concatenate(masses), np.concatenate(intens), prec, z,
                       [float(np.sum(i, dtype=np.float32)) for i in intens])
    small_world.check(ScorerParams(), "edge cases", batch=eb)
    small_world.check(ScorerParams(precursor_tol=Tolerance("da", -5000.0, 5000.0), min_matched_peaks=1), "edge, all peptides",
                      batch=eb)
    # a zero and a tiny peak mass in front of a real spectrum: dividends outside the range the short division by 1e6 is proved for
    # (core.h: FAST_DIV_LO) — prelim_kernel's probe takes the IEEE sequence for that batch of windows (kernels.hip: window_of)
    s5 = slice(int(b.peak_off[5]), int(b.peak_off[6]))
    zm = np.concatenate([np.array([0.0, 1e-30], np.float32), b.masses[s5]])
    zi = np.concatenate([np.array([3.0, 4.0], np.float32), b.intensities[s5]])
    zb = SpectrumBatch([0, len(zm), len(zm) + (s5.stop - s5.start)], np.concatenate([zm, b.masses[s5]]),
                       np.concatenate([zi, b.intensities[s5]]), [float(b.precursor_mz[5])] * 2, [int(b.precursor_charge[5])] * 2,
                       [float(np.sum(zi, dtype=np.float32)), float(np.sum(b.intensities[s5], dtype=np.float32))])
    assert small_world.check(ScorerParams(report_psms=2), "zero and tiny peak masses", batch=zb)[0] >= 2
    # empty batch
    scorer = Scorer(small_world.dev, ScorerParams())
    feats, counts = scorer.score(SpectrumBatch([0], [], [], [], [], []))
    assert feats.shape[0] == 0 and counts.shape[0] == 0


def _check_annotation(world, params, batch, context):
    scorer = Scorer(world.dev, params)
    dbatch = scorer.upload(batch)
    gf, gc = scorer.score_resident(dbatch)
    gf, gc = gf.copy(), gc.copy()
    goff, garr = scorer.annotate(dbatch, gf, gc)
    ooff, oarr = world.orc.annotate(params, batch)
    np.testing.assert_array_equal(goff, ooff, err_msg=f"{context}: PSM offsets")
    for k in ("kinds", "charges", "fragment_ordinals", "intensities", "mz_calculated", "mz_experimental"):
        np.testing.assert_array_equal(garr[k], oarr[k], err_msg=f"{context}: Fragments.{k}")  # f32 values bit-exact
    assert len(garr["kinds"]) == int(goff[-1]) > 0
    return int(goff[-1])


def test_annotate_matches_fragments(small_world):
    """Scorer.annotate_matches (scoring.rs:722-752): kinds / charges / ordinals / intensities / m/z of every matched
    fragment of every reported PSM, in the reference's push order."""
    _check_annotation(small_world, ScorerParams(annotate_matches=True), small_world.batch, "annotate, report 1")
    _check_annotation(small_world, ScorerParams(annotate_matches=True, report_psms=4, max_fragment_charge=3,
                                                precursor_tol=Tolerance("da", -2.0, 2.0)), small_world.batch,
                      "annotate, report 4, fragment charge 3")
    _check_annotation(small_world, ScorerParams(annotate_matches=True, chimera=True, report_psms=3), small_world.batch,
                      "annotate + chimera (peaks removed between PSMs)")


def test_peptides_beyond_1023_residues(gpu_required):
    """The reference puts no limit on a peptide's length (database.rs:96-115: `max_len` is the user's); the production rescoring
    kernel keeps a candidate's longest-run state in 10-bit fields.  A database with longer peptides is scored by the general
    instances (DevScorer::long_runs: the two-register Run form): undigested proteins of 1100-2300 residues, spectra whose most
    intense peaks are a long ladder of consecutive b / y ions far beyond index 1023 — runs, matched counts, Fragments ordinals."""
    from sage_amd.synthetic import _MASS_LUT, PROTON
    rng = np.random.default_rng(91)
    letters = list("ADEFGHILMNQSTVWY")  # no K / R: trypsin finds nothing to cut
    prot = lambda n: "".join(rng.choice(letters, n))  # noqa: E731
    fasta = "".join(f">sp|LONG{i}|LONG{i}\n{prot(n)}\n" for i, n in enumerate((1100, 1300, 1600, 2050, 2300, 2300, 1024, 1023)))
    dbp = DatabaseParameters(bucket_size=1024, enzyme=dict(missed_cleavages=0, min_len=1000, max_len=3000, cleave_at="KR", restrict="P"),
                             peptide_min_mass=500.0, peptide_max_mass=400000.0, static_mods={"C": 57.0215})
    w = World(fasta, dbp, {}, 4, seed=3)  # (the world's own synthetic spectra stop at 2 500 Th: replaced below)
    host = w.host
    assert host.n_peptides == 16 and int(np.diff(host.seq_off.astype(np.int64)).max()) == 2300
    spectra = []
    seq_off = host.seq_off.astype(np.int64)
    for i in range(48):
        pep = int(np.flatnonzero(host.decoy == 0)[i % 8])
        a, b = seq_off[pep], seq_off[pep + 1]
        res = _MASS_LUT[host.seq[a:b]] + host.mods[a:b].astype(np.float64)
        mono = float(host.pep_mono[pep])
        z = int(rng.choice([2, 3, 4]))
        bs = np.cumsum(res)[:-1]
        ys = mono - bs
        L = len(bs)
        lo = int(rng.integers(L // 2, L - 120))  # a ladder of ~100 consecutive ions beyond the middle of the peptide
        n_lad = int(rng.integers(60, 110))
        lad = np.arange(lo, lo + n_lad)
        lad = lad[rng.random(n_lad) < 0.9]  # (with gaps: several runs, one of them the longest)
        mz = np.concatenate([bs[lad] + PROTON, ys[lad[: len(lad) // 3]] + PROTON, (bs[lad[::5]] + 2 * PROTON) / 2.0,
                             rng.uniform(150.0, mono, 60)])
        mz = mz * (1.0 + rng.normal(0.0, 2.0, len(mz)) * 1e-6)
        it = np.concatenate([rng.lognormal(9.0, 0.5, len(mz) - 60), rng.lognormal(6.0, 1.0, 60)])
        order = np.argsort(mz, kind="stable")
        spectra.append(RawSpectrum(mz[order].astype(np.float32), it[order].astype(np.float32),
                                   float(np.float32((mono + z * PROTON) / z)), z, None, scan_start_time=float(i), file_id=0, id=f"scan={i}"))
    sp = SpectrumProcessor(150, False, 0.0)
    w.batch = SpectrumBatch.from_spectra([sp.process(r) for r in spectra])
    tol = Tolerance("da", -200000.0, 200000.0)  # every peptide is a candidate of every spectrum
    n, t = w.check(ScorerParams(precursor_tol=tol, fragment_tol=Tolerance("ppm", -20.0, 20.0), report_psms=3), "long peptides, three PSMs")
    scorer = Scorer(w.dev, ScorerParams(precursor_tol=tol, fragment_tol=Tolerance("ppm", -20.0, 20.0)))
    gf, gc = scorer.score(w.batch)
    best = gf[np.arange(w.batch.n), 0]
    assert int(gc.min()) == 1 and int(best["longest_b"].max()) > 40 and int(best["matched_peaks"].min()) > 50
    w.check(ScorerParams(precursor_tol=tol, fragment_tol=Tolerance("ppm", -20.0, 20.0), chimera=True, report_psms=2, max_fragment_charge=3),
            "long peptides, chimera")
    w.check(ScorerParams(fragment_tol=Tolerance("ppm", -20.0, 20.0)), "long peptides, narrow windows")
    _check_annotation(w, ScorerParams(annotate_matches=True, precursor_tol=tol, fragment_tol=Tolerance("ppm", -20.0, 20.0), report_psms=2),
                      w.batch, "long peptides, annotate")


@pytest.mark.parametrize("low_memory", [False, True])
def test_quick_score_prefilter(small_world, low_memory):
    """Scorer::quick_score (scoring.rs:255-298), both flavours; the low-memory one keeps the report_psms largest
    scores under Score's DERIVED ordering (peptide index first), as heap.rs's `<` / `>` do."""
    for params, ctx in ((ScorerParams(report_psms=2), "narrow"),
                        (ScorerParams(report_psms=3, precursor_tol=Tolerance("da", -3.0, 3.0), min_isotope_err=-1, max_isotope_err=1), "±3 Da x iso"),
                        (ScorerParams(precursor_tol=Tolerance("da", -300.0, 300.0)), "open (large-window kernels)")):
        batch = small_world.batch.subset(np.arange(0, small_world.batch.n, 4 if ctx.startswith("open") else 1))
        scorer = Scorer(small_world.dev, params)
        gk = scorer.quick_score(scorer.upload(batch), low_memory)
        ok = small_world.orc.quick_score(params, batch, low_memory)
        np.testing.assert_array_equal(gk, ok, err_msg=f"quick_score low_memory={low_memory} {ctx}")
        assert gk.sum() > 10


def _experiments_built(world, monkeypatch):
    """The library was built with -DSAGE_HIP_EXPERIMENTS (search_kernel, the fused kernel as a first pass): a default build refuses
    their environment switches when a scorer is created."""
    monkeypatch.setenv("SAGE_HIP_ONE_LAUNCH", "1")
    try:
        Scorer(world.dev, ScorerParams()).close()
        return True
    except L.SageHipError as e:
        assert "SAGE_HIP_EXPERIMENTS" in str(e)
        return False
    finally:
        monkeypatch.delenv("SAGE_HIP_ONE_LAUNCH")


def test_equal_hyperscores_take_the_exact_path(gpu_required, monkeypatch):
    """Order-free trims (the default) are only valid while no two equal hyperscores meet at a reported rank; isoleucine /
    leucine twins (identical masses and fragments) tie exactly, so those spectra must come back through the exact heap
    replay — inside the fused narrow kernel (n_tied), in the retry pass for large windows or the separate kernels (n_retry) —
    and still equal the oracle, whose order is the reference's heap layout."""
    monkeypatch.delenv("SAGE_HIP_EXACT", raising=False)
    fasta = synthetic_fasta(60, seed=17)
    twin = fasta.replace("I", "#").replace("L", "I").replace("#", "L").replace(">sp|SYN", ">sp|TWN")
    params = DatabaseParameters(bucket_size=1024, enzyme=dict(missed_cleavages=1, cleave_at="KR", restrict="P"),
                                static_mods={"C": 57.0215})
    w = World(fasta + twin, params, {}, 300, seed=29)
    # +-10 ppm on this small database: every window holds fewer candidates than trim_hits keeps, no list is ever trimmed, so the
    # order-free list IS the reference's list and the ties are ranked by it without a retry (ST_OK_ORDERED)
    n, t = w.check(ScorerParams(report_psms=2), "I/L twins, windows below k")
    assert t["n_retry"] == 0
    # +-20 Da: a hundred candidates per window, still the narrow kernel — the k-select drops candidates, the heap layout matters
    wide = Tolerance("da", -20.0, 20.0)
    n, t = w.check(ScorerParams(report_psms=2, precursor_tol=wide), "I/L twins, narrow")
    assert t["n_retry"] > 50 and t["n_wide"] == 0
    n, t = w.check(ScorerParams(precursor_tol=Tolerance("da", -200.0, 200.0), report_psms=3), "I/L twins, large windows",
                   batch=w.batch.subset(np.arange(0, 300, 3)))
    assert t["n_retry"] > 10 and t["n_wide"] > 0
    # ONE reported PSM (the default): whichever of the tied candidates wins, its record is what its lane reports anyway — the
    # rescoring wavefront replays bounded_min_heapify from the window counts the first pass kept and lets the earliest tied
    # candidate of the replayed list report, instead of the exact retry pass (n_tied counts those; n_retry what still took the
    # retry pass)
    n, t = w.check(ScorerParams(precursor_tol=wide), "I/L twins, narrow, one PSM: cheap ties")
    assert t["n_tied"] > 50 and t["n_retry"] <= 2 and t["n_wide"] == 0
    n, t = w.check(ScorerParams(precursor_tol=wide, min_matched_peaks=1, fragment_tol=Tolerance("da", -0.5, 0.5)), "one PSM, loose fragments")
    assert t["n_tied"] > 50
    monkeypatch.setenv("SAGE_HIP_NO_FAST_TIES", "1")
    n, t = w.check(ScorerParams(precursor_tol=wide), "I/L twins, narrow, one PSM, cheap ties off")
    assert t["n_retry"] > 50 and t["n_tied"] == 0
    monkeypatch.delenv("SAGE_HIP_NO_FAST_TIES")
    monkeypatch.setenv("SAGE_HIP_WCAP", "128")  # some windows beyond the narrow kernel's counters: those ties take the retry pass
    n, t = w.check(ScorerParams(precursor_tol=wide), "one PSM, mixed routing")
    assert t["n_wide"] > 0 and t["n_tied"] > 0
    monkeypatch.delenv("SAGE_HIP_WCAP")
    # a slot capacity whose window counts no longer fit the LDS the in-line replay stages them in (rows of up to
    # wcap / 2 words against the 5 KB of bitmap + peak table): cheap ties are off for such a scorer, the retry pass settles them
    monkeypatch.setenv("SAGE_HIP_WCAP", "8192")
    n, t = w.check(ScorerParams(precursor_tol=Tolerance("da", -400.0, 400.0)), "one PSM, windows of thousands of slots in the narrow kernel",
                   batch=w.batch.subset(np.arange(0, 300, 2)))
    assert t["n_tied"] == 0 and t["n_retry"] > 20 and t["n_wide"] < 75  # (narrow-kernel windows well beyond 2560 slots among them)
    monkeypatch.delenv("SAGE_HIP_WCAP")
    b0 = w.batch  # no charge annotation: several queries per spectrum — no stored counts, the retry pass settles the tie
    unknown = SpectrumBatch(b0.peak_off, b0.masses, b0.intensities, b0.precursor_mz, np.zeros(b0.n, np.uint8), b0.total_ion_current,
                            b0.isolation_lo, b0.isolation_hi, b0.scan_start_time, b0.inverse_ion_mobility, b0.file_id)
    n, t = w.check(ScorerParams(precursor_tol=wide), "one PSM, unknown charges", batch=unknown)
    assert t["n_retry"] > 20 and t["n_tied"] == 0
    # families of EIGHT peptides with identical masses and fragments (isoleucine / leucine at three positions): eight candidates
    # share the best hyperscore — any number of them is settled the same way
    rng = np.random.default_rng(41)
    fam = []
    for t_ in range(40):
        L_ = int(rng.integers(10, 18))
        body = [str(c) for c in rng.choice(list("ADEFGHMNQSTVWY"), L_)]
        xs = sorted(rng.choice(L_, 3, replace=False).tolist())
        for combo in range(8):
            q = list(body)
            for bit, pos in enumerate(xs):
                q[pos] = "IL"[(combo >> bit) & 1]
            fam.append(f">sp|FAM{t_:02d}{combo}|FAM{t_:02d}{combo}\n{''.join(q)}K\n")
    wf = World("".join(fam), DatabaseParameters(bucket_size=1024, enzyme=dict(missed_cleavages=0, cleave_at="KR", restrict="P")), {}, 150,
               seed=43)
    n, t = wf.check(ScorerParams(precursor_tol=Tolerance("da", -600.0, 600.0)), "one PSM, eight-fold ties")
    assert t["n_tied"] > 20 and t["n_retry"] <= 2 and t["n_wide"] == 0
    monkeypatch.setenv("SAGE_HIP_WAYS", "3")
    big1 = w.batch.subset(np.arange(3 * 8192) % w.batch.n)
    p1 = ScorerParams(precursor_tol=wide)
    scorer = Scorer(w.dev, p1)
    gf, gc = scorer.score_resident(scorer.upload(big1))
    t = scorer.last_timing()
    of, oc, _, _ = w.orc.score(p1, big1)
    assert_features_equal(gf, gc, of, oc, "I/L twins, one PSM, three parts side by side")
    assert t["n_ways"] == 3 and t["n_tied"] > 50
    monkeypatch.delenv("SAGE_HIP_WAYS")
    n, t = w.check(ScorerParams(chimera=True, report_psms=3, precursor_tol=wide), "I/L twins, chimera")
    assert t["n_retry"] > 50
    if _experiments_built(w, monkeypatch):  # (the losing first-pass experiments of DESIGN.md 4.7: builds with -DSAGE_HIP_EXPERIMENTS only)
        monkeypatch.setenv("SAGE_HIP_FUSED", "1")  # the fused narrow kernel: the wavefront goes round again with exact trims
        n, t = w.check(ScorerParams(report_psms=2, precursor_tol=wide), "I/L twins, narrow, fused kernel")
        assert t["n_tied"] > 50 and t["n_retry"] == 0
        n, t = w.check(ScorerParams(chimera=True, report_psms=3, precursor_tol=wide), "I/L twins, chimera, fused kernel")
        assert t["n_tied"] > 50
        n, t = w.check(ScorerParams(min_isotope_err=-1, max_isotope_err=3, report_psms=2, precursor_tol=wide), "I/L twins, isotope errors, fused kernel")
        monkeypatch.delenv("SAGE_HIP_FUSED")
    monkeypatch.setenv("SAGE_HIP_WAYS", "3")  # a resident step in three parts on three streams (needs >= 8192 spectra per part)
    big = w.batch.subset(np.arange(3 * 8192) % w.batch.n)
    p2 = ScorerParams(report_psms=2, precursor_tol=wide)
    scorer = Scorer(w.dev, p2)
    gf, gc = scorer.score_resident(scorer.upload(big))
    t = scorer.last_timing()
    of, oc, _, _ = w.orc.score(p2, big)
    assert_features_equal(gf, gc, of, oc, "I/L twins, narrow, three parts side by side")
    assert t["n_ways"] == 3 and t["n_retry"] > 50
    monkeypatch.delenv("SAGE_HIP_WAYS")
    if _experiments_built(w, monkeypatch):
        monkeypatch.setenv("SAGE_HIP_ONE_LAUNCH", "1")  # preliminary and rescoring workgroups in one launch, handing over through HBM
        n, t = w.check(ScorerParams(report_psms=2, precursor_tol=wide), "I/L twins, narrow, one launch")
        assert t["n_retry"] > 50
        n, t = w.check(ScorerParams(precursor_tol=Tolerance("da", -2.0, 2.0), chimera=True, report_psms=3), "I/L twins, mixed routing, chimera, one launch")
        monkeypatch.delenv("SAGE_HIP_ONE_LAUNCH")
    n, t = w.check(ScorerParams(min_isotope_err=-1, max_isotope_err=3, report_psms=2, precursor_tol=wide), "I/L twins, isotope errors")
    assert t["n_retry"] > 50 and t["n_tied"] == 0
    monkeypatch.setenv("SAGE_HIP_ASSUME_NARROW", "1")  # a batch wrongly taken for narrow-only is scored again with the large-window kernels
    n, t = w.check(ScorerParams(precursor_tol=Tolerance("da", -200.0, 200.0), report_psms=3), "I/L twins, large windows, wrong guess",
                   batch=w.batch.subset(np.arange(0, 300, 3)))
    assert t["n_wide"] > 0
    monkeypatch.delenv("SAGE_HIP_ASSUME_NARROW")
    monkeypatch.setenv("SAGE_HIP_EXACT", "1")  # every trim replays the heap: no retries by construction
    n, t = w.check(ScorerParams(report_psms=2, precursor_tol=wide), "I/L twins, exact mode")
    assert t["n_retry"] == 0 and t["n_tied"] == 0


def test_index_built_on_device(small_world):
    """Parameters::build_from_peptides (database.rs:265-346) on the device: an index generated in HBM from the peptide list
    gives the same PSMs as the host-built one — narrow (both matching variants), tiled large-window and rescoring paths —
    also for a, c, x, z ion kinds and another min_ion_index, and from a peptides-only host database."""
    dev2 = DeviceDatabase(small_world.host, 0, build_on_device=True)
    assert dev2.device_bytes == small_world.dev.device_bytes
    sub = small_world.batch.subset(np.arange(0, small_world.batch.n, 3))
    small_world.check(ScorerParams(), "device-built index, narrow", dev=dev2)
    small_world.check(ScorerParams(precursor_tol=Tolerance("da", -2.0, 2.0), report_psms=3, min_isotope_err=-1, max_isotope_err=1),
                      "device-built index, ±2 Da x iso", dev=dev2)
    small_world.check(ScorerParams(precursor_tol=Tolerance("da", -500.0, 100.0)), "device-built index, open", batch=sub, dev=dev2)
    dev2.close()
    fasta = synthetic_fasta(120, seed=13)
    kw = dict(bucket_size=1024, enzyme=dict(missed_cleavages=2, cleave_at="KR", restrict="P"), static_mods={"C": 57.0215},
              variable_mods={"M": [15.9949], "[": [42.010565]}, ion_kinds=["a", "b", "c", "x", "y", "z"], min_ion_index=1)
    w = World(fasta, DatabaseParameters(**kw), {}, 150, seed=23)
    pep_only = DatabaseParameters(**kw).build(fasta, peptides_only=True)
    assert not pep_only.has_fragments and pep_only.n_peptides == w.host.n_peptides
    dev3 = DeviceDatabase(pep_only, 0)
    w.check(ScorerParams(), "six ion kinds, peptides-only host db", dev=dev3)
    w.check(ScorerParams(precursor_tol=Tolerance("da", -100.0, 100.0), report_psms=2), "six ion kinds, wide", dev=dev3)
    dev3.close()


def _raw_edge_cases(world, rng):
    """Raw spectra that stress SpectrumProcessor::process: isotope envelopes at charges 1-3, duplicated peaks, equal
    intensities, fewer peaks than take_top_n, a single peak, an empty spectrum, unknown precursor charge."""
    raws = synthetic_spectra(world.host, 40, seed=77)
    out = list(raws)
    for k, r in enumerate(raws[:12]):
        mz, it = [r.mz], [r.intensity]
        for z in (1, 2, 3):  # C13 satellites below the monoisotopic intensity, plus one ABOVE it (not merged)
            pick = rng.choice(len(r.mz), 12, replace=False)
            mz.append((r.mz[pick] + np.float32(1.00335 / z)).astype(np.float32))
            it.append((r.intensity[pick] * np.float32(0.6 if k % 2 else 1.4)).astype(np.float32))
        mz, it = np.concatenate(mz), np.concatenate(it)
        if k % 3 == 0:  # exact duplicates of (m/z, intensity) and ties in intensity
            mz, it = np.concatenate([mz, mz[:20]]), np.concatenate([it, it[:20]])
            it[30:60] = it[30]
        o = np.argsort(mz, kind="stable")
        out.append(RawSpectrum(mz[o], it[o], r.precursor_mz, None if k % 4 == 0 else r.precursor_charge, r.isolation_window,
                               r.scan_start_time, None, 0, f"edge={k}"))
    out.append(RawSpectrum(raws[0].mz[:7], raws[0].intensity[:7], 500.0, 2, None, 0.0, None, 0, "few"))
    out.append(RawSpectrum(np.array([300.5], np.float32), np.array([10.0], np.float32), 500.0, 2, None, 0.0, None, 0, "one"))
    out.append(RawSpectrum(np.zeros(0, np.float32), np.zeros(0, np.float32), 500.0, 2, None, 0.0, None, 0, "empty"))
    return out


@pytest.mark.parametrize("deisotope,top_n", [(True, 150), (True, 40), (False, 150), (False, 25)])
def test_device_spectrum_processing(small_world, deisotope, top_n):
    """SpectrumProcessor::process on the device (spectrum.rs:179-227, 279-412) against the CPU oracle: masses,
    intensities and total ion current bit-exact, then the same PSMs from the resident processed batch."""
    raws = _raw_edge_cases(small_world, np.random.default_rng(5))
    scorer = Scorer(small_world.dev, ScorerParams())
    dbatch, npk = scorer.process_upload(RawBatch(raws), take_top_n=top_n, deisotope=deisotope, min_deisotope_mz=0.0, min_peaks=0)
    off, m, it, tic = dbatch.download()
    for i, r in enumerate(raws):
        om, oi, otic = oracle_lib.process_ms2(top_n, deisotope, 0.0, r.mz, r.intensity, r.precursor_charge)
        a, b = int(off[i]), int(off[i + 1])
        assert b - a == len(om) == npk[i], f"spectrum {r.id}: {b - a} peaks vs oracle {len(om)}"
        np.testing.assert_array_equal(m[a:b], om, err_msg=f"masses of {r.id}")
        np.testing.assert_array_equal(it[a:b], oi, err_msg=f"intensities of {r.id}")
        assert np.float32(tic[i]) == np.float32(otic), f"TIC of {r.id}"
    # scoring the device-processed batch == scoring the host-processed spectra (min_peaks filter of runner.rs:313)
    dbatch2, _ = scorer.process_upload(RawBatch(raws), take_top_n=top_n, deisotope=deisotope, min_peaks=15)
    gf, gc = scorer.score_resident(dbatch2)
    gf, gc = gf.copy(), gc.copy()
    sp = SpectrumProcessor(top_n, deisotope, 0.0)
    proc = [sp.process(r) for r in raws]
    keep = [i for i, p_ in enumerate(proc) if len(p_.masses) >= 15]
    hb = SpectrumBatch.from_spectra([proc[i] for i in keep])
    of, oc, _, _ = small_world.orc.score(ScorerParams(), hb)
    assert gc.sum() == oc.sum() and all(gc[i] == 0 for i in range(len(raws)) if i not in keep)
    sub_f, sub_c = gf[keep], gc[keep]
    sub_f = sub_f.copy()
    sub_f["spec_index"] = np.where(np.arange(sub_f.shape[1])[None, :] < sub_c[:, None], np.arange(len(keep))[:, None], sub_f["spec_index"])
    assert_features_equal(sub_f, sub_c, of, oc, "device-processed batch")


@pytest.mark.parametrize("deisotope", [True, False])
def test_large_raw_spectra_and_a_thousand_peaks(small_world, deisotope):
    """Raw spectra far beyond the preprocessing kernel's LDS share (20 000 and 3 000 raw peaks: the global-workspace instance of
    process_kernel) next to ordinary ones, max_peaks = 1000 (input.rs:366 is user-set): processed peaks bit-exact, then the
    search over those 1000-peak spectra against the oracle (spectrum.rs:279-412, scoring.rs:300-309)."""
    rng = np.random.default_rng(41)
    raws = list(synthetic_spectra(small_world.host, 12, seed=78))
    for k, n_extra in enumerate((20000, 3000, 2100)):
        r = raws[k]
        mz = np.concatenate([r.mz, rng.uniform(120.0, 1900.0, n_extra).astype(np.float32)])
        it = np.concatenate([r.intensity * np.float32(50.0), rng.gamma(2.0, 20.0, n_extra).astype(np.float32)])
        o = np.argsort(mz, kind="stable")
        raws.append(RawSpectrum(mz[o], it[o], r.precursor_mz, r.precursor_charge, r.isolation_window, r.scan_start_time, None, 0, f"big={n_extra}"))
    top_n = 1000
    scorer = Scorer(small_world.dev, ScorerParams())
    dbatch, npk = scorer.process_upload(RawBatch(raws), take_top_n=top_n, deisotope=deisotope, min_deisotope_mz=0.0, min_peaks=0)
    off, m, it, tic = dbatch.download()
    for i, r in enumerate(raws):
        om, oi, otic = oracle_lib.process_ms2(top_n, deisotope, 0.0, r.mz, r.intensity, r.precursor_charge)
        a, b = int(off[i]), int(off[i + 1])
        assert b - a == len(om) == npk[i], f"spectrum {r.id}: {b - a} peaks vs oracle {len(om)}"
        np.testing.assert_array_equal(m[a:b], om, err_msg=f"masses of {r.id}")
        np.testing.assert_array_equal(it[a:b], oi, err_msg=f"intensities of {r.id}")
        assert np.float32(tic[i]) == np.float32(otic), f"TIC of {r.id}"
    assert npk.max() == top_n
    gf, gc = scorer.score_resident(dbatch)
    sp = SpectrumProcessor(top_n, deisotope, 0.0)
    hb = SpectrumBatch.from_spectra([sp.process(r) for r in raws])
    of, oc, _, _ = small_world.orc.score(ScorerParams(), hb)
    assert assert_features_equal(gf, gc, of, oc, "1000-peak spectra") >= 10
    # ... and through the large-window kernels (peaks x fragment charges in the count kernel's LDS)
    params = ScorerParams(precursor_tol=Tolerance("da", -300.0, 300.0), report_psms=2)
    scorer2 = Scorer(small_world.dev, params)
    gf, gc = scorer2.score(hb)
    of, oc, _, _ = small_world.orc.score(params, hb)
    assert_features_equal(gf, gc, of, oc, "1000-peak spectra, large windows")


def test_spectra_of_five_thousand_peaks(small_world):
    """`max_peaks` is the user's (sage-cli input.rs:366).  5 400 peaks x fragment charges 1..3 = 16 200 (peak, charge) windows per
    spectrum — twice what rounds 1-4 could stage in LDS (they answered SAGE_HIP_ERR_UNSUPPORTED): the preliminary kernel probes
    (peak masses only in LDS), the rescoring kernels take more than the default 64 KB of a compute unit's LDS, and the large-window
    count kernel keeps the windows in global memory (tile_count_wing_kernel).  Narrow, open, chimera + unknown charge, annotation."""
    rng = np.random.default_rng(43)
    raws = []
    for k, r in enumerate(synthetic_spectra(small_world.host, 10, seed=79, charges=((4, 1.0),))):
        n_extra = 9000 if k < 8 else 200  # (two ordinary spectra in the batch as well)
        mz = np.concatenate([r.mz, rng.uniform(120.0, 1900.0, n_extra).astype(np.float32)])
        it = np.concatenate([r.intensity * np.float32(50.0), rng.gamma(2.0, 20.0, n_extra).astype(np.float32)])
        o = np.argsort(mz, kind="stable")
        raws.append(RawSpectrum(mz[o], it[o], r.precursor_mz, r.precursor_charge, r.isolation_window, r.scan_start_time, None, 0, f"big{k}"))
    top_n = 5400
    sp = SpectrumProcessor(top_n, False, 0.0)
    hb = SpectrumBatch.from_spectra([sp.process(r) for r in raws])
    assert int(np.diff(hb.peak_off.astype(np.int64)).max()) == top_n
    w = small_world
    n, t = w.check(ScorerParams(max_fragment_charge=3), "5400 peaks, narrow", batch=hb, every=3)
    assert n >= 8 and t["n_wide"] == 0
    n, t = w.check(ScorerParams(max_fragment_charge=3, precursor_tol=Tolerance("da", -300.0, 300.0), report_psms=2), "5400 peaks, large windows",
                   batch=hb, every=3)
    assert t["n_wide"] > 0
    unknown = SpectrumBatch(hb.peak_off, hb.masses, hb.intensities, hb.precursor_mz, np.zeros(hb.n, np.uint8), hb.total_ion_current)
    w.check(ScorerParams(chimera=True, report_psms=2, precursor_tol=Tolerance("da", -2.0, 2.0)), "5400 peaks, unknown charge, chimera",
            batch=unknown, every=3)
    _check_annotation(w, ScorerParams(annotate_matches=True, max_fragment_charge=3), hb, "5400 peaks, annotate")
    # the device's own preprocessing feeds the same kernels
    scorer = Scorer(w.dev, ScorerParams(max_fragment_charge=3))
    dbatch, npk = scorer.process_upload(RawBatch(raws), take_top_n=top_n, deisotope=False, min_deisotope_mz=0.0, min_peaks=0)
    gf, gc = scorer.score_resident(dbatch)
    of, oc, _, _ = w.orc.score(ScorerParams(max_fragment_charge=3), hb)
    assert_features_equal(gf, gc, of, oc, "5400 peaks, device preprocessing")


def test_error_paths(small_world):
    with pytest.raises(L.SageHipError):
        Scorer(small_world.dev, ScorerParams(report_psms=0))
    with pytest.raises(L.SageHipError):
        Scorer(small_world.dev, ScorerParams(min_isotope_err=2, max_isotope_err=1))


def test_streaming_pipeline_chunks(small_world, monkeypatch):
    """sage_hip_score_batch as a pipeline over chunks (here 64 spectra instead of 65536, so ten chunks rotate through the four
    input slots and, without large windows, the two compute lanes): pageable and page-locked inputs, page-locked and pageable outputs, spec_index in the caller's numbering; narrow,
    isotope-folded / unknown-charge, mixed narrow / tiled, chimera."""
    monkeypatch.setenv("SAGE_HIP_CHUNK", "64")
    b = small_world.batch
    unknown = SpectrumBatch(b.peak_off, b.masses, b.intensities, b.precursor_mz, np.zeros(b.n, np.uint8), b.total_ion_current,
                            b.isolation_lo, b.isolation_hi, b.scan_start_time, b.inverse_ion_mobility, b.file_id)
    for params, batch, ctx in ((ScorerParams(), b, "narrow"),
                               (ScorerParams(min_isotope_err=-1, max_isotope_err=2, report_psms=3), unknown, "iso x charges"),
                               (ScorerParams(precursor_tol=Tolerance("da", -2.0, 2.0), report_psms=2), b, "mixed routing"),
                               (ScorerParams(chimera=True, report_psms=3), b, "chimera")):
        scorer = Scorer(small_world.dev, params)
        of, oc, _, _ = small_world.orc.score(params, batch)
        locked = batch.page_locked()
        for inp, pinned_out, how in ((batch, True, "pageable in"), (locked, True, "page-locked in"), (batch, False, "pageable out")):
            gf, gc = scorer.score(inp, pinned_out=pinned_out)
            n = assert_features_equal(gf, gc, of, oc, f"pipeline {ctx}, {how}")
            valid = np.arange(gf.shape[1])[None, :] < gc[:, None]
            assert np.array_equal(gf["spec_index"][valid], np.broadcast_to(np.arange(batch.n)[:, None], gf.shape)[valid])
        t = scorer.last_timing()
        assert n > 100 and t["n_launches"] >= 10 * 3  # ten chunks: two kernels + the retry pass each, more with the large-window path
    # the estimate says "no large windows" and is wrong: the first chunk that notices is scored again with the large-window
    # kernels while its neighbours are in flight on the other lane, and the chunks after it take the large-window route at once
    monkeypatch.setenv("SAGE_HIP_ASSUME_NARROW", "1")
    params = ScorerParams(precursor_tol=Tolerance("da", -200.0, 200.0), report_psms=2)
    scorer = Scorer(small_world.dev, params)
    sub = b.subset(np.arange(0, 3 * (b.n // 3), 3)[:200])
    of, oc, _, _ = small_world.orc.score(params, sub)
    for inp, pinned_out in ((sub, True), (sub.page_locked(), True), (sub, False)):
        assert_features_equal(*scorer.score(inp, pinned_out=pinned_out), of, oc, "pipeline, large windows, wrong guess")
    assert scorer.last_timing()["n_wide"] > 0
    monkeypatch.delenv("SAGE_HIP_ASSUME_NARROW")
    monkeypatch.setenv("SAGE_HIP_ONE_LANE", "1")  # all chunks on one compute stream
    scorer = Scorer(small_world.dev, ScorerParams(report_psms=3))
    of, oc, _, _ = small_world.orc.score(ScorerParams(report_psms=3), b)
    assert_features_equal(*scorer.score(b), of, oc, "pipeline, one lane")
    monkeypatch.delenv("SAGE_HIP_ONE_LANE")
    # a chunk boundary that leaves a last chunk of one spectrum, and a batch smaller than a chunk
    scorer = Scorer(small_world.dev, ScorerParams())
    for m in (65, 64, 3, 1):
        sub = b.subset(np.arange(m))
        of, oc, _, _ = small_world.orc.score(ScorerParams(), sub)
        assert_features_equal(*scorer.score(sub), of, oc, f"pipeline, {m} spectra")


def test_scorer_shared_between_threads_and_cloned(small_world):
    """`&Scorer` is shared by every rayon worker (scoring.rs:300, runner.rs:311-325): one handle called from several host
    threads at once (calls serialise), and clones of it scoring concurrently — all results equal the oracle's."""
    import threading
    params = ScorerParams(report_psms=2)
    scorer = Scorer(small_world.dev, params)
    of, oc, _, _ = small_world.orc.score(params, small_world.batch)
    parts = [small_world.batch.subset(np.arange(k, small_world.batch.n, 4)) for k in range(4)]
    refs = [small_world.orc.score(params, p)[:2] for p in parts]
    errors = []

    def work(sc, k, reps):
        try:
            for _ in range(reps):
                gf, gc = sc.score(parts[k], pinned_out=False)
                assert_features_equal(gf, gc, refs[k][0], refs[k][1], f"thread {k}")
        except Exception as e:  # noqa: BLE001 (reported below)
            errors.append(repr(e))

    for scorers in ([scorer] * 4, [scorer.clone() for _ in range(4)]):
        ts = [threading.Thread(target=work, args=(scorers[k], k, 5)) for k in range(4)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not errors, errors
    assert_features_equal(*scorer.score(small_world.batch), of, oc, "after the threads")


def test_arena_exhaustion_splits_the_batch(small_world, monkeypatch):
    """A large-window candidate arena too small for the chunk (1 MiB instead of ~1 GiB): the chunk is scored again in halves
    until its pieces fit — same PSMs, no error."""
    monkeypatch.setenv("SAGE_HIP_ARENA_MB", "1")
    params = ScorerParams(precursor_tol=Tolerance("da", -500.0, 100.0))
    sub = small_world.batch.subset(np.arange(0, small_world.batch.n, 5))
    scorer = Scorer(small_world.dev, params)
    of, oc, _, _ = small_world.orc.score(params, sub)
    gf, gc = scorer.score(sub)
    assert_features_equal(gf, gc, of, oc, "arena split")
    assert scorer.last_timing()["n_launches"] > 12  # more than one piece


def test_precursor_window_table_finds_the_same_windows(small_world, monkeypatch):
    """The narrow kernel finds a precursor window's partition points through a position table over the peptide masses
    (DevDbView::pep_lut, 1/128 Da bins) instead of the full five-level search.  Same candidates, bit for bit: the preliminary
    lists against the oracle with the table; a database built WITHOUT the table (SAGE_HIP_NO_PEP_LUT=1) gives identical
    records; windows that leave the table's domain (an absurd precursor m/z beyond the heaviest peptide, a zero one, Da
    tolerances wider than the first bin) take the full search and agree too."""
    w = small_world
    for params, ctx in ((ScorerParams(), "default"), (ScorerParams(precursor_tol=Tolerance("ppm", -5.0, 5.0)), "5 ppm"),
                        (ScorerParams(precursor_tol=Tolerance("da", -3.0, 1.0), report_psms=2), "-3/+1 Da"),
                        (ScorerParams(precursor_tol=Tolerance("da", -0.004, 0.004)), "half a bin"),
                        (ScorerParams(precursor_tol=Tolerance("ppm", -20.0, 20.0), min_isotope_err=-1, max_isotope_err=3), "isotope errors")):
        w.check(params, "table: " + ctx)
    monkeypatch.setenv("SAGE_HIP_NO_PEP_LUT", "1")
    plain = DeviceDatabase(w.host, 0)
    monkeypatch.delenv("SAGE_HIP_NO_PEP_LUT")
    for params in (ScorerParams(), ScorerParams(precursor_tol=Tolerance("da", -3.0, 1.0), report_psms=2)):
        a, b = Scorer(w.dev, params), Scorer(plain, params)
        fa, ca = a.score(w.batch)
        fb, cb = b.score(w.batch)
        np.testing.assert_array_equal(ca, cb)
        valid = np.arange(fa.shape[1])[None, :] < ca[:, None]
        assert fa[valid].tobytes() == fb[valid].tobytes()
    # spectra whose windows leave the table: precursor m/z 0, tiny, and far beyond the heaviest peptide
    b0 = w.batch.subset(np.arange(12))
    mz = b0.precursor_mz.copy()
    mz[0], mz[1], mz[2], mz[3] = 0.0, 1.0e-3, 1.0e5, 3.0e7
    odd = SpectrumBatch(b0.peak_off, b0.masses, b0.intensities, mz, b0.precursor_charge, b0.total_ion_current, b0.isolation_lo,
                        b0.isolation_hi, b0.scan_start_time, b0.inverse_ion_mobility, b0.file_id)
    w.check(ScorerParams(), "table: windows outside its domain", batch=odd)
    w.check(ScorerParams(precursor_tol=Tolerance("da", -5.0, 5.0)), "table: windows across the table's first bins", batch=odd)
