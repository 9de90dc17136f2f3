"""Shared helpers for the GPU-vs-oracle parity tests."""
import numpy as np

EXACT_FIELDS = ["spec_index", "peptide_idx", "rank", "label", "charge", "matched_peaks", "longest_b", "longest_y",
                "scored_candidates", "peptide_len", "missed_cleavages", "file_id",
                # f32 values produced by the same IEEE operations in the same order on both sides
                "expmass", "calcmass", "rt", "ims", "delta_mass", "isotope_error", "average_ppm", "longest_y_pct",
                "matched_intensity_pct", "ms2_intensity"]
# f64 values that go through ln().  The product computes a correctly rounded ln (sage_amd/csrc/crlog.h) and the same IEEE f64
# operations in the same order as the reference around it, so against the oracle in its correctly-rounded mode (libquadmath,
# oracle_lib's default) these fields are held to EQUALITY.  Against the oracle with the platform libm (oracle_lib.LogMode(0):
# glibc 2.35 here, 0.52 ulp) the last bit may differ for ~0.01 % of the arguments: REL_TOL, and at least MIN_EQUAL of the values
# equal.  (The north star's tolerance for hyperscore is 1e-4 relative.)
REL_FIELDS = ["hyperscore", "delta_next", "delta_best", "poisson"]
REL_TOL = 1e-15
MIN_EQUAL = 0.995
# differences of two ~equal hyperscores amplify the ulp error: absolute tolerance on the deltas (platform-libm mode only)
DELTA_ABS = 1e-13


def assert_features_equal(gf, gc, of, oc, context="", rel_tol=None, exact_f64=None):
    import oracle_lib
    if exact_f64 is None:
        exact_f64 = rel_tol is None and oracle_lib.log_mode() == 1
    rel = REL_TOL if rel_tol is None else rel_tol
    np.testing.assert_array_equal(gc, oc, err_msg=f"{context}: PSM counts differ")
    n, r = gf.shape
    mask = np.arange(r)[None, :] < gc[:, None]
    g, o = gf[mask], of[mask]
    for f in EXACT_FIELDS + (REL_FIELDS if exact_f64 else []):
        a, b = g[f], o[f]
        if a.dtype.kind == "f":
            same = (a == b) | (np.isnan(a) & np.isnan(b))
        else:
            same = a == b
        if not np.all(same):
            bad = np.flatnonzero(~same)[:5]
            raise AssertionError(f"{context}: field {f} differs at {bad}: gpu={a[bad]!r} oracle={b[bad]!r} "
                                 f"(spec {g['spec_index'][bad]})")
    for f in ([] if exact_f64 else REL_FIELDS):
        a, b = g[f], o[f]
        both_inf = np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))
        tol = rel * np.maximum(np.abs(b), 1.0) + (max(DELTA_ABS, 100 * rel) if f.startswith("delta") else 0.0)
        ok = both_inf | (np.abs(a - b) <= tol)
        if not np.all(ok):
            bad = np.flatnonzero(~ok)[:5]
            raise AssertionError(f"{context}: field {f} differs at {bad}: gpu={a[bad]} oracle={b[bad]}")
        if rel_tol is None and len(a) >= 1000:
            frac = float(np.mean((a == b) | both_inf))
            assert frac >= MIN_EQUAL, f"{context}: field {f}: only {frac:.4f} of the values equal the platform libm's"
    return int(mask.sum())


def assert_initial_hits_equal(scorer, dbatch, orc, params, batch, context="", every=1):
    packed, ln, mp, sc = scorer.initial_hits(dbatch)
    for i in range(0, batch.n, every):
        op, omp, osc = orc.initial_hits(params, batch, i)
        assert ln[i] == len(op), f"{context}: spectrum {i}: list length {ln[i]} vs oracle {len(op)}"
        if not np.array_equal(packed[i, :ln[i]], op):
            raise AssertionError(f"{context}: spectrum {i}: preliminary list differs\n gpu   ={packed[i, :ln[i]]}\n oracle={op}")
        assert mp[i] == omp and sc[i] == osc, f"{context}: spectrum {i}: totals ({mp[i]},{sc[i]}) vs ({omp},{osc})"
