"""Holds the C++ oracle (oracle/sage_oracle.cpp) to a SECOND, independent restatement of the reference's scoring
(tests/second_reading.py: numpy / np.float32 straight loops written from scoring.rs, spectrum.rs, heap.rs, ion_series.rs and
database.rs alone — no fragment index, brute-force matching).  CPU only.

The reference's tests pin `matched_peaks == 21` on one spectrum and nothing else of Scorer::score's output; everything the GPU
parity suite proves is "product == oracle".  This file is the second pin on the oracle itself: preliminary lists INCLUDING heap
order, every integer and f32 Feature field bit for bit, the f64 fields to 1e-12 — on the reference's fixture and on ≥ 500
synthetic spectra over narrow, isotope-error, unknown-charge, chimeric wide-window, OpenMS-score, six-ion-kind and
k-select-binding searches.
"""
import numpy as np
import pytest

import oracle_lib
import second_reading as SR
from sage_amd import _lib as L
from sage_amd.api import DatabaseParameters, ScorerParams, SpectrumBatch, SpectrumProcessor, Tolerance
from sage_amd.synthetic import synthetic_fasta, synthetic_spectra
from test_oracle_golden import c1_batch, integration_scorer

INT_FIELDS = ["peptide_idx", "rank", "label", "charge", "matched_peaks", "longest_b", "longest_y", "scored_candidates",
              "peptide_len", "missed_cleavages", "file_id"]
F32_FIELDS = ["expmass", "calcmass", "rt", "ims", "delta_mass", "isotope_error", "average_ppm", "longest_y_pct",
              "matched_intensity_pct", "ms2_intensity"]
F64_FIELDS = ["hyperscore", "delta_next", "delta_best", "poisson"]


def pack(pre):
    m, p, z, iso = pre
    return (m << 48) | (p << 16) | (z << 8) | (iso + 128)


def hold_oracle_to_second_reading(orc, dbp, params, batch, f64_tol=1e-12, context=""):
    """Every spectrum of `batch`: oracle Scorer::score / initial_hits vs the second reading.  Returns (#PSMs, #spectra whose
    k-select had something to cut)."""
    kinds = [L.ION_KINDS[k] for k in (dbp.ion_kinds if dbp.ion_kinds is not None else ["b", "y"])]
    sr = SR.SecondScorer(SR.Peptides(orc.arrays()), kinds, 2 if dbp.min_ion_index is None else dbp.min_ion_index, params)
    ofeat, ocnt, _, _ = orc.score(params, batch)
    n_psm = 0
    for i in range(batch.n):
        feats, hits = sr.score(SR.spectrum_of(batch, i))
        ctx = f"{context} spectrum {i}"
        op, omp, osc = orc.initial_hits(params, batch, i)
        mine = np.array([pack(h) for h in hits[2]], dtype=np.uint64)
        assert len(mine) == len(op), f"{ctx}: preliminary list has {len(op)} entries, second reading {len(mine)}"
        assert np.array_equal(mine, op), f"{ctx}: preliminary list (heap order included) differs\n oracle={op}\n second={mine}"
        assert (omp, osc) == (hits[0], hits[1]), f"{ctx}: totals oracle ({omp}, {osc}) vs second ({hits[0]}, {hits[1]})"
        assert ocnt[i] == len(feats), f"{ctx}: oracle reports {ocnt[i]} PSMs, second reading {len(feats)}"
        for r, f in enumerate(feats):
            o = ofeat[i, r]
            for k in INT_FIELDS:
                assert int(o[k]) == int(f[k]), f"{ctx} rank {r + 1}: {k}: oracle {o[k]} vs second {f[k]}"
            for k in F32_FIELDS:
                a, b = np.float32(o[k]), np.float32(f[k])
                assert a.view(np.uint32) == b.view(np.uint32) or (np.isnan(a) and np.isnan(b)), \
                    f"{ctx} rank {r + 1}: {k}: oracle {a!r} vs second {b!r}"
            for k in F64_FIELDS:
                a, b = float(o[k]), float(f[k])
                if np.isinf(a) or np.isinf(b):
                    assert a == b, f"{ctx} rank {r + 1}: {k}: oracle {a} vs second {b}"
                else:
                    # deltas are differences of two ~equal hyperscores: the tolerance is relative to the hyperscores
                    scale = max(abs(b), abs(float(f["hyperscore"])), 1.0)
                    assert abs(a - b) <= f64_tol * scale, f"{ctx} rank {r + 1}: {k}: oracle {a!r} vs second {b!r}"
            n_psm += 1
    return n_psm


def test_second_reading_reproduces_the_references_known_answer():
    """crates/sage-cli/tests/integration.rs:47-49 through the second reading alone: one PSM, matched_peaks == 21 — and the
    oracle agrees with it on every other field of that PSM."""
    d, batch = c1_batch(oracle_lib.process_ms2)
    dbp = DatabaseParameters()
    orc = oracle_lib.OracleDb.build(d["fasta"], dbp)
    params = integration_scorer()
    sr = SR.SecondScorer(SR.Peptides(orc.arrays()), [SR.B, SR.Y], 2, params)
    feats, hits = sr.score(SR.spectrum_of(batch, 0))
    assert len(feats) == 1 and feats[0]["matched_peaks"] == 21
    assert hold_oracle_to_second_reading(orc, dbp, params, batch, context="C1") == 1
    # the config file's own parameters (tests/config.json: chimera false, report_psms 1 … but isotope errors -1..3, charge 2..4)
    for kw in (dict(report_psms=3, min_matched_peaks=1), dict(chimera=True, report_psms=2, min_matched_peaks=2),
               dict(score_type="OpenMSHyperScore", max_fragment_charge=None, report_psms=2, min_matched_peaks=1)):
        p = integration_scorer()
        for k, v in kw.items():
            setattr(p, k, v)
        hold_oracle_to_second_reading(orc, dbp, p, batch, f64_tol=1e-6 if p.score_type != "SageHyperScore" else 1e-12,
                                      context=f"C1 {kw}")


def _world(n_proteins, seed, dbkw, n_spectra, spec_seed, speckw, quantize=None):
    fasta = synthetic_fasta(n_proteins, seed=seed)
    dbp = DatabaseParameters(**dbkw)
    prod = dbp.build(fasta)
    orc = oracle_lib.OracleDb.build(fasta, dbp)
    assert orc.n_peptides == prod.n_peptides
    sp = SpectrumProcessor(150, True, 0.0)
    spectra = [sp.process(r) for r in synthetic_spectra(prod, n_spectra, spec_seed, **speckw)]
    spectra = [s for s in spectra if len(s.masses) >= 15]
    if quantize:
        # intensities on a coarse grid: several peaks of EQUAL intensity inside one fragment window, where
        # select_most_intense_peak's `>=` (spectrum.rs:153: the LAST of the equals wins) is observable
        for s in spectra:
            s.intensities = (np.ceil(s.intensities / np.float32(quantize)) * np.float32(quantize)).astype(np.float32)
    return dbp, orc, SpectrumBatch.from_spectra(spectra)


_TRYPTIC = dict(bucket_size=1024, enzyme=dict(missed_cleavages=1, min_len=5, max_len=50, cleave_at="KR", restrict="P"),
                static_mods={"C": 57.0215}, variable_mods={"M": [15.9949], "[": [42.010565]}, max_variable_mods=2)

CASES = {
    # C2 / C3's search: ±10 ppm, charge annotated, one PSM
    "narrow": (dict(n_proteins=1500, seed=11, dbkw=_TRYPTIC, n_spectra=180, spec_seed=101, speckw=dict(varmod_frac=0.15)),
               dict(), 1e-12),
    # tests/config.json's fan-out: isotope errors -1..3, half of the precursors without a charge (three charge states searched)
    "isotope_errors": (dict(n_proteins=600, seed=12, dbkw=_TRYPTIC, n_spectra=60, spec_seed=102, speckw=dict()),
                       dict(min_isotope_err=-1, max_isotope_err=3, precursor_tol=Tolerance("ppm", -50.0, 50.0), report_psms=2,
                            min_matched_peaks=2), 1e-12),
    "unknown_charge": (dict(n_proteins=600, seed=13, dbkw=_TRYPTIC, n_spectra=60, spec_seed=103,
                            speckw=dict(annotate_charge=False)),
                       dict(min_isotope_err=0, max_isotope_err=1, report_psms=3, max_fragment_charge=2), 1e-12),
    # C5's search: co-isolated peptides, no charge, wide window, chimera rounds
    "chimera_wide_window": (dict(n_proteins=300, seed=14, dbkw=_TRYPTIC, n_spectra=70, spec_seed=104,
                                 speckw=dict(chimeric=3, isolation_half_width=6.0, annotate_charge=False)),
                            dict(wide_window=True, chimera=True, report_psms=3), 1e-12),
    "chimera_narrow": (dict(n_proteins=1000, seed=15, dbkw=_TRYPTIC, n_spectra=40, spec_seed=105, speckw=dict(chimeric=2)),
                       dict(chimera=True, report_psms=3, min_matched_peaks=3), 1e-12),
    # the other ScoreType (f32 ln_1p: numpy's log1pf and the oracle's may differ by an f32 ulp → 1e-6)
    "openms_score": (dict(n_proteins=1000, seed=16, dbkw=_TRYPTIC, n_spectra=60, spec_seed=106, speckw=dict()),
                     dict(score_type="OpenMSHyperScore", report_psms=4, min_matched_peaks=1), 1e-6),
    # all six ion kinds, min_ion_index 1, no decoys, Da fragment tolerance, asymmetric precursor tolerance
    "abcxyz_da": (dict(n_proteins=1500, seed=17, dbkw=dict(bucket_size=256, ion_kinds=["a", "b", "c", "x", "y", "z"],
                                                         min_ion_index=1, generate_decoys=False,
                                                         enzyme=dict(missed_cleavages=0, cleave_at="KR", restrict="P")),
                       n_spectra=40, spec_seed=107, speckw=dict()),
                  dict(fragment_tol=Tolerance("da", -0.02, 0.01), precursor_tol=Tolerance("ppm", -20.0, 5.0), report_psms=2), 1e-12),
    # windows of hundreds of candidates: trim_hits has something to cut and the heap's layout decides tie order
    "k_select_binds": (dict(n_proteins=600, seed=18, dbkw=_TRYPTIC, n_spectra=40, spec_seed=108, speckw=dict()),
                       dict(precursor_tol=Tolerance("da", -3.0, 3.0), report_psms=5, min_matched_peaks=1), 1e-12),
    # half-Dalton fragment windows over quantised intensities: ties inside select_most_intense_peak, several peaks per window
    "equal_intensities": (dict(n_proteins=600, seed=20, dbkw=_TRYPTIC, n_spectra=40, spec_seed=110, speckw=dict(),
                               quantize=20000.0),
                          dict(fragment_tol=Tolerance("da", -0.6, 0.6), report_psms=2, chimera=True, min_matched_peaks=2), 1e-12),
    # report_psms beyond 25: k = 2·report_psms (scoring.rs:323-326), pct tolerance
    "many_psms_pct": (dict(n_proteins=600, seed=19, dbkw=_TRYPTIC, n_spectra=20, spec_seed=109, speckw=dict()),
                      dict(precursor_tol=Tolerance("da", -2.0, 2.0), fragment_tol=Tolerance("pct", -0.001, 0.001), report_psms=40,
                           min_matched_peaks=1), 1e-12),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_equals_second_reading(name):
    world, skw, tol = CASES[name]
    dbp, orc, batch = _world(**world)
    params = ScorerParams(**skw)
    n_psm = hold_oracle_to_second_reading(orc, dbp, params, batch, f64_tol=tol, context=name)
    assert n_psm >= batch.n // 3, f"{name}: only {n_psm} PSMs over {batch.n} spectra — the case does not exercise the scoring"


def test_the_cases_cover_five_hundred_spectra():
    assert sum(c[0]["n_spectra"] for c in CASES.values()) >= 500
