"""results.sage.tsv / matched_fragments.sage.tsv writers (host side).

Column order and number formatting follow the reference byte for byte where the hot path owns the value:
sage-cli/src/runner.rs:687-780 (serialize_feature), :782-828 (serialize_fragments), :830-935 (headers).  Integers are
written with `itoa`, floats with `ryu` (shortest digits that round-trip, Rust `ryu::Buffer::format`), reproduced by
`ryu_f32` / `ryu_f64` below.  sage_discriminant_score, posterior_error, spectrum_q, peptide_q and protein_q come from the
device rescoring (sage_hip_rescore), aligned_rt / predicted_rt / delta_rt_model / predicted_mobility / delta_mobility from
sage_hip_predict_rt when `predict_rt` is on (the default, input.rs:372); without it, and for the protein-group columns
nothing here computes, the defaults Feature gets in build_features (scoring.rs:576-592): predicted_rt 0.0, aligned_rt = rt,
delta_rt_model / delta_ims_model 0.999, protein_group_q 1.0.  `results.sage.pin` follows runner.rs:938-1135.
"""
import ctypes
import re
from typing import List, Optional, Sequence

import numpy as np

HEADERS = ["psm_id", "peptide", "proteins", "protein_groups", "num_proteins", "num_protein_groups", "filename", "scannr",
           "rank", "label", "expmass", "calcmass", "charge", "peptide_len", "missed_cleavages", "semi_enzymatic",
           "isotope_error", "precursor_ppm", "fragment_ppm", "hyperscore", "delta_next", "delta_best", "rt", "aligned_rt",
           "predicted_rt", "delta_rt_model", "ion_mobility", "predicted_mobility", "delta_mobility", "matched_peaks",
           "longest_b", "longest_y", "longest_y_pct", "matched_intensity_pct", "scored_candidates", "poisson",
           "sage_discriminant_score", "posterior_error", "spectrum_q", "peptide_q", "protein_q", "protein_group_q",
           "ms2_intensity"]
FRAGMENT_HEADERS = ["psm_id", "fragment_type", "fragment_ordinals", "fragment_charge", "fragment_mz_calculated",
                    "fragment_mz_experimental", "fragment_intensity"]
ION_NAMES = "abcxyz"


def _ryu(x, f32: bool) -> str:
    """ryu::Buffer::format: shortest round-trip digits, laid out as ryu's pretty printer does (ryu/src/pretty/mod.rs:
    plain decimals while the decimal point stays within 16 (f64) / 13 (f32) digits and the value is >= 1e-5 (f64) /
    1e-6 (f32); otherwise d.ddde[-]x)."""
    x = np.float32(x) if f32 else np.float64(x)
    if np.isnan(x):
        return "NaN"
    if np.isinf(x):
        return "inf" if x > 0 else "-inf"
    if x == 0:
        return "-0.0" if np.signbit(x) else "0.0"
    sci = np.format_float_scientific(x, unique=True, trim="-", exp_digits=1)  # e.g. '1.2345e+3', '-5e-7'
    sign = "-" if sci[0] == "-" else ""
    mant, exp = sci.lstrip("-").split("e")
    digits = mant.replace(".", "")
    e10 = int(exp)
    length = len(digits)
    kk = e10 + 1            # 10^(kk-1) <= |x| < 10^kk
    k = kk - length         # x = digits * 10^k
    hi = 13 if f32 else 16
    lo = -6 if f32 else -5
    if 0 <= k and kk <= hi:
        return sign + digits + "0" * k + ".0"
    if 0 < kk <= hi:
        return sign + digits[:kk] + "." + digits[kk:]
    if lo < kk <= 0:
        return sign + "0." + "0" * (-kk) + digits
    if length == 1:
        return sign + digits + "e" + str(kk - 1)
    return sign + digits[0] + "." + digits[1:] + "e" + str(kk - 1)


def ryu_f32(x) -> str:
    return _ryu(x, True)


def ryu_f64(x) -> str:
    return _ryu(x, False)


DEFAULT_POST = dict(discriminant_score=0.0, posterior_error=1.0, spectrum_q=1.0, peptide_q=1.0, protein_q=1.0)


def feature_row(psm_id: int, f, db, filename: str, scannr: str, post: Optional[dict] = None) -> List[str]:
    """serialize_feature (runner.rs:687-780) for one SageFeature record `f` (numpy void of FEATURE_DTYPE); `post` = the
    rescoring outputs of this PSM (defaults of scoring.rs:576-592 when the rescoring did not run)."""
    post = dict(DEFAULT_POST, **(post or {}))
    pep = int(f["peptide_idx"])
    num_proteins, semi = db.peptide_info(pep)
    rt = f["rt"]
    aligned_rt = post.get("aligned_rt", rt)  # Feature defaults without the predict_rt block: scoring.rs:576-592
    predicted_rt, delta_rt = post.get("predicted_rt", 0.0), post.get("delta_rt_model", 0.999)
    predicted_ims, delta_ims = post.get("predicted_ims", 0.0), post.get("delta_ims_model", 0.999)
    return [
        str(psm_id), db.peptide_string(pep), db.peptide_proteins(pep), "", str(num_proteins), "0", filename, scannr,
        str(int(f["rank"])), str(int(f["label"])), ryu_f32(f["expmass"]), ryu_f32(f["calcmass"]), str(int(f["charge"])),
        str(int(f["peptide_len"])), str(int(f["missed_cleavages"])), str(semi), ryu_f32(f["isotope_error"]),
        ryu_f32(f["delta_mass"]), ryu_f32(f["average_ppm"]), ryu_f64(f["hyperscore"]), ryu_f64(f["delta_next"]),
        ryu_f64(f["delta_best"]), ryu_f32(rt), ryu_f32(aligned_rt), ryu_f32(predicted_rt), ryu_f32(delta_rt), ryu_f32(f["ims"]),
        ryu_f32(predicted_ims), ryu_f32(delta_ims), str(int(f["matched_peaks"])), str(int(f["longest_b"])), str(int(f["longest_y"])),
        ryu_f32(f["longest_y_pct"]), ryu_f32(f["matched_intensity_pct"]), str(int(f["scored_candidates"])),
        ryu_f64(f["poisson"]), ryu_f32(post["discriminant_score"]), ryu_f32(post["posterior_error"]),
        ryu_f32(post["spectrum_q"]), ryu_f32(post["peptide_q"]), ryu_f32(post["protein_q"]), ryu_f32(1.0),
        ryu_f32(f["ms2_intensity"]),
    ]


def write_features(path: str, rows: Sequence[List[str]]) -> None:
    """write_features (runner.rs:830-905): tab-separated, header first."""
    with open(path, "w", newline="") as fh:
        fh.write("\t".join(HEADERS) + "\n")
        for r in rows:
            fh.write("\t".join(r) + "\n")


def fragment_rows(psm_id: int, lo: int, hi: int, arr) -> List[List[str]]:
    """serialize_fragments (runner.rs:782-828) for entries [lo, hi) of the flat Fragments arrays."""
    return [[str(psm_id), ION_NAMES[int(arr["kinds"][j])], str(int(arr["fragment_ordinals"][j])), str(int(arr["charges"][j])),
             ryu_f32(arr["mz_calculated"][j]), ryu_f32(arr["mz_experimental"][j]), ryu_f32(arr["intensities"][j])]
            for j in range(lo, hi)]


def write_fragments(path: str, rows: Sequence[List[str]]) -> None:
    with open(path, "w", newline="") as fh:
        fh.write("\t".join(FRAGMENT_HEADERS) + "\n")
        for r in rows:
            fh.write("\t".join(r) + "\n")


PIN_HEADERS = ["SpecId", "Label", "ScanNr", "ExpMass", "CalcMass", "FileName", "retentiontime", "ion_mobility", "rank", "z=2",
               "z=3", "z=4", "z=5", "z=6", "z=other", "peptide_len", "missed_cleavages", "semi_enzymatic", "isotope_error",
               "ln(precursor_ppm)", "fragment_ppm", "ln(hyperscore)", "ln(delta_next)", "ln(delta_best)", "aligned_rt",
               "predicted_rt", "sqrt(delta_rt_model)", "predicted_mobility", "sqrt(delta_mobility)", "matched_peaks", "longest_b",
               "longest_y", "longest_y_pct", "ln(matched_intensity_pct)", "scored_candidates", "ln(-poisson)", "posterior_error",
               "Peptide", "Proteins"]
_SCAN_RE = re.compile(r"scan=(\d+)")
_libm = ctypes.CDLL("libm.so.6")
_libm.log1pf.restype = ctypes.c_float
_libm.log1pf.argtypes = [ctypes.c_float]
_libm.log1p.restype = ctypes.c_double
_libm.log1p.argtypes = [ctypes.c_double]


def _ln1p_f32(x) -> np.float32:  # f32::ln_1p == libm log1pf
    return np.float32(_libm.log1pf(float(np.float32(x))))


def _ln1p_f64(x) -> np.float64:
    return np.float64(_libm.log1p(float(x)))


def pin_row(psm_id: int, f, db, filename: str, spec_id: str, post: Optional[dict] = None) -> List[str]:
    """serialize_pin (runner.rs:938-1084): percolator input, log / sqrt transformed feature columns."""
    post = dict(DEFAULT_POST, **(post or {}))
    pep = int(f["peptide_idx"])
    _, semi = db.peptide_info(pep)
    caps = _SCAN_RE.findall(spec_id)
    scannr = caps[-1] if caps else spec_id
    z = int(f["charge"])
    rt = f["rt"]
    aligned_rt = post.get("aligned_rt", rt)
    predicted_rt, predicted_ims = post.get("predicted_rt", 0.0), post.get("predicted_ims", 0.0)
    d = np.float32(post.get("delta_rt_model", 0.999))
    delta_rt = np.float32(0.001) if d < np.float32(0.001) else (np.float32(1.0) if d > np.float32(1.0) else d)  # .clamp(0.001, 1.0)
    delta_ims = post.get("delta_ims_model", 0.999)
    return [
        str(psm_id), str(int(f["label"])), scannr, ryu_f32(f["expmass"]), ryu_f32(f["calcmass"]), filename, ryu_f32(rt),
        ryu_f32(f["ims"]), str(int(f["rank"])), str(int(z == 2)), str(int(z == 3)), str(int(z == 4)), str(int(z == 5)),
        str(int(z == 6)), str(z if (z < 2 or z > 6) else 0), str(int(f["peptide_len"])), str(int(f["missed_cleavages"])),
        str(semi), ryu_f32(f["isotope_error"]), ryu_f32(_ln1p_f32(np.abs(np.float32(f["delta_mass"])))),
        ryu_f32(f["average_ppm"]), ryu_f64(_ln1p_f64(f["hyperscore"])), ryu_f64(_ln1p_f64(f["delta_next"])),
        ryu_f64(_ln1p_f64(f["delta_best"])), ryu_f32(aligned_rt), ryu_f32(predicted_rt), ryu_f32(np.sqrt(delta_rt, dtype=np.float32)),
        ryu_f32(predicted_ims), ryu_f32(delta_ims), str(int(f["matched_peaks"])), str(int(f["longest_b"])), str(int(f["longest_y"])),
        ryu_f32(f["longest_y_pct"]), ryu_f32(_ln1p_f32(f["matched_intensity_pct"])), str(int(f["scored_candidates"])),
        ryu_f64(_ln1p_f64(-np.float64(f["poisson"]))), ryu_f32(post["posterior_error"]), db.peptide_string(pep),
        db.peptide_proteins(pep),
    ]


def write_pin(path: str, rows: Sequence[List[str]]) -> None:
    """write_pin (runner.rs:1086-1135)."""
    with open(path, "w", newline="") as fh:
        fh.write("\t".join(PIN_HEADERS) + "\n")
        for r in rows:
            fh.write("\t".join(r) + "\n")


def write_results_native(path: str, fmt: str, db, features, order, psm_ids, filenames, spec_ids, post=None) -> None:
    """results.sage.tsv (fmt "tsv") / results.sage.pin (fmt "pin") through the C++ writer (sage_hip_write_results): the same
    bytes as feature_row / pin_row + write_features / write_pin, without a Python loop per PSM.  `post`: RescoreResult-like
    and / or RtPrediction-like objects (attributes named like SagePostColumns), or None."""
    import ctypes as C

    from . import _lib as L

    f = np.ascontiguousarray(features, dtype=L.FEATURE_DTYPE).reshape(-1)
    n = len(f)
    order_a = None if order is None else np.ascontiguousarray(order, dtype=np.uint64)
    ids = np.ascontiguousarray(psm_ids, dtype=np.uint64)
    assert len(ids) == n and len(spec_ids) == n and (order_a is None or len(order_a) == n)
    names = (C.c_char_p * max(len(filenames), 1))(*[s.encode() for s in filenames])
    specs = (C.c_char_p * max(n, 1))(*[s.encode() for s in spec_ids])
    cols = L.SagePostColumns()
    keep = []
    for source in (post if isinstance(post, (list, tuple)) else [post]):
        for k, _ in L.SagePostColumns._fields_:
            a = getattr(source, k, None) if source is not None else None
            if a is not None:
                a = np.ascontiguousarray(a, dtype=np.float32)
                assert len(a) == n
                keep.append(a)
                setattr(cols, k, L.as_ptr(a, C.c_float))
    L.check(L.load().sage_hip_write_results(path.encode(), {"tsv": 0, "pin": 1}[fmt], db._h, f.ctypes.data, n,
                                            None if order_a is None else L.as_ptr(order_a, C.c_uint64), L.as_ptr(ids, C.c_uint64),
                                            names, len(filenames), specs, C.byref(cols)))
