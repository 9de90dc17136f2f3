"""Shared helpers for the GPU-vs-oracle parity tests."""
import numpy as np

EXACT_FIELDS = ["spec_index", "peptide_idx", "rank", "label", "charge", "matched_peaks", "longest_b", "longest_y",
                "scored_candidates", "peptide_len", "missed_cleavages", "file_id",
                # f32 values produced by the same IEEE operations in the same order on both sides
                "expmass", "calcmass", "rt", "ims", "delta_mass", "isotope_error", "average_ppm", "longest_y_pct",
                "matched_intensity_pct", "ms2_intensity"]
# f64 values that go through ln(): device libm vs glibc may differ in the last ulp.  The north star's
# tolerance for hyperscore is 1e-4 relative; we hold the device to 1e-12.
REL_FIELDS = ["hyperscore", "delta_next", "delta_best", "poisson"]
REL_TOL = 1e-12
# differences of two ~equal hyperscores amplify the ulp error: absolute tolerance on the deltas
DELTA_ABS = 1e-10


def assert_features_equal(gf, gc, of, oc, context="", rel_tol=None):
    rel = REL_TOL if rel_tol is None else rel_tol
    np.testing.assert_array_equal(gc, oc, err_msg=f"{context}: PSM counts differ")
    n, r = gf.shape
    mask = np.arange(r)[None, :] < gc[:, None]
    g, o = gf[mask], of[mask]
    for f in EXACT_FIELDS:
        a, b = g[f], o[f]
        if a.dtype.kind == "f":
            same = (a == b) | (np.isnan(a) & np.isnan(b))
        else:
            same = a == b
        if not np.all(same):
            bad = np.flatnonzero(~same)[:5]
            raise AssertionError(f"{context}: field {f} differs at {bad}: gpu={a[bad]} oracle={b[bad]} "
                                 f"(spec {g['spec_index'][bad]})")
    for f in REL_FIELDS:
        a, b = g[f], o[f]
        both_inf = np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))
        tol = rel * np.maximum(np.abs(b), 1.0) + (max(DELTA_ABS, 100 * rel) if f.startswith("delta") else 0.0)
        ok = both_inf | (np.abs(a - b) <= tol)
        if not np.all(ok):
            bad = np.flatnonzero(~ok)[:5]
            raise AssertionError(f"{context}: field {f} differs at {bad}: gpu={a[bad]} oracle={b[bad]}")
    return int(mask.sum())


def assert_initial_hits_equal(scorer, dbatch, orc, params, batch, context="", every=1):
    packed, ln, mp, sc = scorer.initial_hits(dbatch)
    for i in range(0, batch.n, every):
        op, omp, osc = orc.initial_hits(params, batch, i)
        assert ln[i] == len(op), f"{context}: spectrum {i}: list length {ln[i]} vs oracle {len(op)}"
        if not np.array_equal(packed[i, :ln[i]], op):
            raise AssertionError(f"{context}: spectrum {i}: preliminary list differs\n gpu   ={packed[i, :ln[i]]}\n oracle={op}")
        assert mp[i] == omp and sc[i] == osc, f"{context}: spectrum {i}: totals ({mp[i]},{sc[i]}) vs ({omp},{osc})"
