"""ctypes wrapper over oracle/_build/liboracle.so — TEST INFRASTRUCTURE (the CPU restatement of the
reference).  Struct layouts are shared with sage_amd._lib because oracle_capi.cpp mirrors sage_hip.h."""
import ctypes as C
import os
import subprocess

import numpy as np

from sage_amd import _lib as L
from sage_amd.api import DatabaseParameters, ScorerParams, SpectrumBatch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "_build", "liboracle.so")
# the same sources built for speed (oracle/Makefile: FASTFLAGS); bit-identical outputs; bench.py's cpu_baseline times this one
LIB_FAST = os.path.join(ORACLE_DIR, "_build", "liboracle_fast.so")
SELFTEST = os.path.join(ORACLE_DIR, "_build", "oracle_selftest")


class OrcWork(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("queries", "page_searches", "pages", "scanned", "hits", "peaks", "rescored",
                                          "rescored_residues", "reported", "algorithmic_bytes")]


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


_libs = {}
_kind = "check"


class use:
    """with oracle_lib.use("fast"): objects created inside bind to the performance build of the oracle."""

    def __init__(self, kind):
        self.kind = kind

    def __enter__(self):
        global _kind
        self.prev, _kind = _kind, self.kind

    def __exit__(self, *a):
        global _kind
        _kind = self.prev


def load(kind=None):
    kind = kind or _kind
    if kind in _libs:
        return _libs[kind]
    LIB = LIB_FAST if kind == "fast" else globals()["LIB"]
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("sage_oracle.cpp", "sage_oracle.hpp", "oracle_capi.cpp", "rescore_oracle.cpp",
                                                "selftest.cpp")]
    if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs if os.path.exists(s)):
        build()
    lib = C.CDLL(LIB)
    vp = C.c_void_p
    lib.orc_db_build.restype = vp
    lib.orc_db_build.argtypes = [C.c_char_p, C.POINTER(L.SageDbParams)]
    lib.orc_db_from_arrays.restype = vp
    lib.orc_db_from_arrays.argtypes = [L.c_u32_p, L.c_float_p, C.c_uint64, L.c_float_p, C.c_uint64, C.c_uint64,
                                       L.c_float_p, L.c_u64_p, L.c_u8_p, L.c_float_p, L.c_float_p, L.c_float_p,
                                       L.c_u8_p, L.c_u8_p, C.c_uint64, L.c_u8_p, C.c_uint32]
    lib.orc_db_free.argtypes = [vp]
    for f in ("orc_db_num_peptides", "orc_db_num_fragments", "orc_db_num_buckets", "orc_db_bucket_size",
              "orc_db_total_residues"):
        getattr(lib, f).restype = C.c_uint64
        getattr(lib, f).argtypes = [vp]
    lib.orc_db_copy_fragments.argtypes = [vp, L.c_u32_p, L.c_float_p]
    lib.orc_db_copy_min_value.argtypes = [vp, L.c_float_p]
    lib.orc_db_copy_peptides.argtypes = [vp, L.c_float_p, L.c_u8_p, L.c_u8_p, L.c_float_p, L.c_float_p, L.c_u64_p,
                                         L.c_u8_p, L.c_float_p]
    lib.orc_db_peptide_strings.restype = C.c_uint64
    lib.orc_db_peptide_strings.argtypes = [vp, C.c_char_p, C.c_uint64]
    lib.orc_db_peptide_proteins.restype = C.c_uint64
    lib.orc_db_peptide_proteins.argtypes = [vp, C.c_uint64, C.c_char_p, C.c_uint64]
    lib.orc_db_page_search.restype = C.c_uint64
    lib.orc_db_page_search.argtypes = [vp, C.c_float, L.SageTolerance, L.SageTolerance, C.c_float, L.c_u64_p,
                                       C.c_uint64, L.c_u64_p, L.c_u64_p]
    lib.orc_process_ms2.restype = C.c_uint64
    lib.orc_process_ms2.argtypes = [C.c_uint64, C.c_int, C.c_float, L.c_float_p, L.c_float_p, C.c_uint64, C.c_uint8,
                                    L.c_float_p, L.c_float_p, L.c_float_p]
    lib.orc_score_batch.restype = C.c_double
    lib.orc_score_batch.argtypes = [vp, C.POINTER(L.SageScorerParams), C.POINTER(L.SageSpectrumBatch), vp, L.c_u32_p,
                                    C.c_int, C.POINTER(OrcWork)]
    lib.orc_initial_hits.restype = C.c_uint64
    lib.orc_initial_hits.argtypes = [vp, C.POINTER(L.SageScorerParams), C.POINTER(L.SageSpectrumBatch), C.c_uint32,
                                     C.POINTER(C.c_uint16), L.c_u32_p, L.c_u8_p, C.POINTER(C.c_int8), C.c_uint64,
                                     L.c_u64_p, L.c_u64_p]
    lib.orc_brute_force.restype = C.c_uint64
    lib.orc_brute_force.argtypes = [vp, C.POINTER(L.SageScorerParams), C.POINTER(L.SageSpectrumBatch), C.c_uint32,
                                    C.c_uint8, C.c_int8, L.c_u32_p, L.c_u32_p, C.POINTER(C.c_double), C.c_uint64]
    lib.orc_annotate_batch.restype = C.c_uint64
    lib.orc_annotate_batch.argtypes = [vp, C.POINTER(L.SageScorerParams), C.POINTER(L.SageSpectrumBatch), L.c_u64_p, L.c_u8_p,
                                       C.POINTER(C.c_int32), C.POINTER(C.c_int32), L.c_float_p, L.c_float_p, L.c_float_p,
                                       C.c_uint64]
    lib.orc_quick_score.argtypes = [vp, C.POINTER(L.SageScorerParams), C.POINTER(L.SageSpectrumBatch), C.c_int, L.c_u8_p]
    lib.orc_tol_bounds.argtypes = [L.SageTolerance, C.c_float, L.c_float_p, L.c_float_p]
    lib.orc_max_threads.restype = C.c_int
    lib.orc_db_build_chunk.restype = vp
    lib.orc_db_build_chunk.argtypes = [C.c_char_p, C.POINTER(L.SageDbParams), C.c_uint64, C.c_uint64]
    lib.orc_fasta_num_targets.restype = C.c_uint64
    lib.orc_fasta_num_targets.argtypes = [C.c_char_p, C.POINTER(L.SageDbParams)]
    lib.orc_prefilter_chunk_size.restype = C.c_uint64
    lib.orc_prefilter_chunk_size.argtypes = [C.c_char_p, C.POINTER(L.SageDbParams), C.c_uint64]
    lib.orc_db_merge_kept.restype = vp
    lib.orc_db_merge_kept.argtypes = [C.POINTER(vp), C.POINTER(L.c_u8_p), C.c_uint32, C.POINTER(L.SageDbParams)]
    dp = C.POINTER(C.c_double)
    # f64::ln of the reference: 0 = the platform libm, 1 = correctly rounded through libquadmath (oracle/sage_oracle.cpp: ln).
    # The checker build compares in mode 1 — the product's contract (sage_amd/csrc/crlog.h), bit for bit; the performance build
    # that bench.py times as cpu_baseline keeps the platform libm, the reference's arithmetic on this host.
    lib.orc_set_log_mode.argtypes = [C.c_int]
    lib.orc_get_log_mode.restype = C.c_int
    lib.orc_ln_batch.argtypes = [C.c_int, dp, C.c_uint64, dp]
    lib.orc_set_log_mode(0 if kind == "fast" else 1)
    lib.orc_rescore_mode.argtypes = [C.c_int]
    lib.orc_lda_train.restype = C.c_int
    lib.orc_lda_train.argtypes = [dp, C.c_uint64, C.c_uint64, L.c_u8_p, dp]
    lib.orc_gauss_solve.restype = C.c_int
    lib.orc_gauss_solve.argtypes = [dp, dp, C.c_uint64, dp]
    lib.orc_kde.argtypes = [dp, L.c_u8_p, C.c_uint64, C.c_int, C.c_uint64, C.c_double, dp, dp, dp, C.c_uint64, dp]
    lib.orc_rescore.restype = C.c_int
    lib.orc_rescore.argtypes = [vp, C.c_uint64, C.c_int, C.c_float, C.c_float, L.c_float_p, L.c_float_p, L.c_float_p,
                                L.c_u32_p, C.c_uint32, L.c_u32_p, C.c_uint32, L.c_float_p, L.c_float_p, L.c_float_p,
                                L.c_float_p, L.c_float_p, L.c_u32_p, L.c_u64_p, dp, dp]
    lib.orc_linreg_fit.restype = C.c_int
    lib.orc_linreg_fit.argtypes = [dp, dp, L.c_u8_p, C.c_uint64, C.c_uint64, dp, dp]
    lib.orc_rt_embed.argtypes = [L.c_u8_p, C.c_uint64, C.c_float, dp]
    lib.orc_im_embed.argtypes = [L.c_u8_p, C.c_uint64, C.c_float, C.c_uint8, dp]
    lib.orc_predict_rt.argtypes = [vp, C.c_uint64, C.c_uint32, L.c_u64_p, L.c_u8_p, L.c_float_p] + [L.c_float_p] * 7 + \
        [C.POINTER(C.c_int32), dp]
    _libs[kind] = lib
    return lib


class OracleDb:
    """An oracle IndexedDatabase, built from FASTA (its own builder) or adopted from flat arrays."""

    def __init__(self, handle):
        self.h = handle
        lib = self.lib = load()
        self.n_peptides = int(lib.orc_db_num_peptides(self.h))
        self.n_fragments = int(lib.orc_db_num_fragments(self.h))
        self.n_buckets = int(lib.orc_db_num_buckets(self.h))
        self.bucket_size = int(lib.orc_db_bucket_size(self.h))

    @staticmethod
    def build(fasta_text: str, params: DatabaseParameters) -> "OracleDb":
        p, keep = params.to_c()
        return OracleDb(C.c_void_p(load().orc_db_build(fasta_text.encode(), C.byref(p))))

    @staticmethod
    def build_chunk(fasta_text: str, params: DatabaseParameters, first: int, count: int) -> "OracleDb":
        p, keep = params.to_c()
        return OracleDb(C.c_void_p(load().orc_db_build_chunk(fasta_text.encode(), C.byref(p), first, count)))

    @staticmethod
    def merge_kept(chunks, keeps, params: DatabaseParameters) -> "OracleDb":
        p, keep = params.to_c()
        masks = [np.ascontiguousarray(k, dtype=np.uint8) for k in keeps]
        hs = (C.c_void_p * max(len(chunks), 1))(*[c.h for c in chunks])
        ms = (L.c_u8_p * max(len(chunks), 1))(*[L.as_ptr(m, C.c_uint8) for m in masks])
        return OracleDb(C.c_void_p(load().orc_db_merge_kept(hs, ms, len(chunks), C.byref(p))))

    @staticmethod
    def from_product(db) -> "OracleDb":
        """Adopt the arrays of a sage_amd.IndexedDatabase (same inputs for both legs of a comparison)."""
        lib = load()
        fp = np.ascontiguousarray(db.fragments["peptide_index"])
        fm = np.ascontiguousarray(db.fragments["fragment_mz"])
        h = lib.orc_db_from_arrays(L.as_ptr(fp, C.c_uint32), L.as_ptr(fm, C.c_float), len(fp),
                                   L.as_ptr(db.min_value, C.c_float), len(db.min_value), db.bucket_size,
                                   L.as_ptr(db.pep_mono, C.c_float), L.as_ptr(db.seq_off, C.c_uint64),
                                   L.as_ptr(db.seq, C.c_uint8), L.as_ptr(db.mods, C.c_float),
                                   L.as_ptr(db.nterm, C.c_float), L.as_ptr(db.cterm, C.c_float),
                                   L.as_ptr(db.decoy, C.c_uint8), L.as_ptr(db.missed_cleavages, C.c_uint8),
                                   db.n_peptides, L.as_ptr(db.ion_kinds, C.c_uint8), len(db.ion_kinds))
        return OracleDb(C.c_void_p(h))

    def arrays(self):
        lib = self.lib
        nf, np_, nb = self.n_fragments, self.n_peptides, self.n_buckets
        tot = int(lib.orc_db_total_residues(self.h))
        out = dict(frag_pep=np.zeros(nf, np.uint32), frag_mz=np.zeros(nf, np.float32), min_value=np.zeros(nb, np.float32),
                   pep_mono=np.zeros(np_, np.float32), decoy=np.zeros(np_, np.uint8), missed=np.zeros(np_, np.uint8),
                   nterm=np.zeros(np_, np.float32), cterm=np.zeros(np_, np.float32), seq_off=np.zeros(np_ + 1, np.uint64),
                   seq=np.zeros(tot, np.uint8), mods=np.zeros(tot, np.float32))
        lib.orc_db_copy_fragments(self.h, L.as_ptr(out["frag_pep"], C.c_uint32), L.as_ptr(out["frag_mz"], C.c_float))
        lib.orc_db_copy_min_value(self.h, L.as_ptr(out["min_value"], C.c_float))
        lib.orc_db_copy_peptides(self.h, L.as_ptr(out["pep_mono"], C.c_float), L.as_ptr(out["decoy"], C.c_uint8),
                                 L.as_ptr(out["missed"], C.c_uint8), L.as_ptr(out["nterm"], C.c_float),
                                 L.as_ptr(out["cterm"], C.c_float), L.as_ptr(out["seq_off"], C.c_uint64),
                                 L.as_ptr(out["seq"], C.c_uint8), L.as_ptr(out["mods"], C.c_float))
        return out

    def peptide_strings(self):
        lib = self.lib
        n = lib.orc_db_peptide_strings(self.h, None, 0)
        buf = C.create_string_buffer(int(n))
        lib.orc_db_peptide_strings(self.h, buf, n)
        s = buf.value.decode()
        return s.split("\n")[:-1] if s else []

    def peptide_proteins(self, i):
        lib = self.lib
        n = lib.orc_db_peptide_proteins(self.h, i, None, 0)
        buf = C.create_string_buffer(int(n))
        lib.orc_db_peptide_proteins(self.h, i, buf, n)
        return buf.value.decode()

    def page_search(self, precursor_mass, ptol, ftol, mass, cap=1 << 20):
        lib = self.lib
        idx = np.zeros(cap, np.uint64)
        lo, hi = C.c_uint64(), C.c_uint64()
        n = lib.orc_db_page_search(self.h, precursor_mass, ptol.to_c(), ftol.to_c(), mass, L.as_ptr(idx, C.c_uint64),
                                   cap, C.byref(lo), C.byref(hi))
        return idx[:int(n)].copy(), int(lo.value), int(hi.value)

    def score(self, params: ScorerParams, batch: SpectrumBatch, threads: int = 0, work: bool = False):
        """Scorer::score for each spectrum.  Returns (features[n, report], counts[n], elapsed_ms, work|None)."""
        lib = self.lib
        cp = params.to_c()
        cb = batch.to_c()
        feats = np.zeros(batch.n * params.report_psms, dtype=L.FEATURE_DTYPE)
        counts = np.zeros(batch.n, np.uint32)
        w = OrcWork()
        ms = lib.orc_score_batch(self.h, C.byref(cp), C.byref(cb), feats.ctypes.data_as(C.c_void_p),
                                 L.as_ptr(counts, C.c_uint32), threads, C.byref(w) if work else None)
        wd = {k: getattr(w, k) for k, _ in OrcWork._fields_} if work else None
        return feats.reshape(batch.n, params.report_psms), counts, float(ms), wd

    def annotate(self, params: ScorerParams, batch: SpectrumBatch, cap: int = 1 << 22):
        """Fragments (scoring.rs:152-161) of every reported PSM: (psm_off[n*report+1], dict of flat arrays)."""
        lib = self.lib
        cp = params.to_c(); cb = batch.to_c()
        off = np.zeros(batch.n * params.report_psms + 1, np.uint64)
        out = dict(kinds=np.zeros(cap, np.uint8), charges=np.zeros(cap, np.int32), fragment_ordinals=np.zeros(cap, np.int32),
                   intensities=np.zeros(cap, np.float32), mz_calculated=np.zeros(cap, np.float32),
                   mz_experimental=np.zeros(cap, np.float32))
        n = int(lib.orc_annotate_batch(self.h, C.byref(cp), C.byref(cb), L.as_ptr(off, C.c_uint64), L.as_ptr(out["kinds"], C.c_uint8),
                                       out["charges"].ctypes.data_as(C.POINTER(C.c_int32)),
                                       out["fragment_ordinals"].ctypes.data_as(C.POINTER(C.c_int32)),
                                       L.as_ptr(out["intensities"], C.c_float), L.as_ptr(out["mz_calculated"], C.c_float),
                                       L.as_ptr(out["mz_experimental"], C.c_float), cap))
        assert n <= cap
        return off, {k: v[:n].copy() for k, v in out.items()}

    def quick_score(self, params: ScorerParams, batch: SpectrumBatch, prefilter_low_memory: bool):
        lib = self.lib
        cp = params.to_c(); cb = batch.to_c()
        keep = np.zeros(self.n_peptides, np.uint8)
        lib.orc_quick_score(self.h, C.byref(cp), C.byref(cb), int(prefilter_low_memory), L.as_ptr(keep, C.c_uint8))
        return keep

    def initial_hits(self, params: ScorerParams, batch: SpectrumBatch, i: int):
        lib = self.lib
        cap = 4096
        m = np.zeros(cap, np.uint16); p = np.zeros(cap, np.uint32); z = np.zeros(cap, np.uint8); iso = np.zeros(cap, np.int8)
        mp, sc = C.c_uint64(), C.c_uint64()
        cp = params.to_c(); cb = batch.to_c()
        n = int(lib.orc_initial_hits(self.h, C.byref(cp), C.byref(cb), i, m.ctypes.data_as(C.POINTER(C.c_uint16)),
                                     L.as_ptr(p, C.c_uint32), L.as_ptr(z, C.c_uint8),
                                     iso.ctypes.data_as(C.POINTER(C.c_int8)), cap, C.byref(mp), C.byref(sc)))
        packed = (m[:n].astype(np.uint64) << np.uint64(48)) | (p[:n].astype(np.uint64) << np.uint64(16)) | \
                 (z[:n].astype(np.uint64) << np.uint64(8)) | ((iso[:n].astype(np.int64) + 128).astype(np.uint64))
        return packed, int(mp.value), int(sc.value)

    def brute_force(self, params: ScorerParams, batch: SpectrumBatch, i: int, charge: int, iso: int, cap=1 << 16):
        lib = self.lib
        p = np.zeros(cap, np.uint32); m = np.zeros(cap, np.uint32); h = np.zeros(cap, np.float64)
        cp = params.to_c(); cb = batch.to_c()
        n = int(lib.orc_brute_force(self.h, C.byref(cp), C.byref(cb), i, charge, iso, L.as_ptr(p, C.c_uint32),
                                    L.as_ptr(m, C.c_uint32), h.ctypes.data_as(C.POINTER(C.c_double)), cap))
        return p[:n], m[:n], h[:n]

    def close(self):
        if self.h:
            self.lib.orc_db_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def process_ms2(take_top_n, deisotope, min_deisotope_mz, mz, intensity, precursor_charge):
    lib = load()
    mz = np.ascontiguousarray(mz, np.float32); it = np.ascontiguousarray(intensity, np.float32)
    n = len(mz)
    om = np.zeros(max(n, 1), np.float32); oi = np.zeros(max(n, 1), np.float32); tic = C.c_float()
    k = int(lib.orc_process_ms2(take_top_n, int(deisotope), min_deisotope_mz, L.as_ptr(mz, C.c_float),
                                L.as_ptr(it, C.c_float), n, precursor_charge or 0, L.as_ptr(om, C.c_float),
                                L.as_ptr(oi, C.c_float), C.byref(tic)))
    return om[:k].copy(), oi[:k].copy(), float(np.float32(tic.value))


def tol_bounds(tol, center):
    lo, hi = C.c_float(), C.c_float()
    load().orc_tol_bounds(tol.to_c(), center, C.byref(lo), C.byref(hi))
    return np.float32(lo.value), np.float32(hi.value)


# ---- post-search rescoring (oracle/rescore_oracle.cpp) ----------------------------------------------------------------
def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class _mode:
    """det=False: the reference's own summation order and the platform libm (the restatement proper).  det=True: the order and
    the elementary functions the device evaluates (sage_amd/csrc/detmath.h) — what the GPU is held to bit for bit."""

    def __init__(self, det):
        self.det = det

    def __enter__(self):
        load().orc_rescore_mode(1 if self.det else 0)

    def __exit__(self, *a):
        load().orc_rescore_mode(0)


def lda_train(rows, decoy, det=False):
    """LinearDiscriminantAnalysis::train (linear_discriminant.rs:57-127) -> coefficients or None"""
    rows = np.ascontiguousarray(rows, dtype=np.float64)
    decoy = np.ascontiguousarray(decoy, dtype=np.uint8)
    coef = np.zeros(rows.shape[1])
    with _mode(det):
        ok = load().orc_lda_train(_dptr(rows), rows.shape[0], rows.shape[1], L.as_ptr(decoy, C.c_uint8), _dptr(coef))
    return coef if ok else None


def gauss_solve(left, right):
    left = np.ascontiguousarray(left, dtype=np.float64)
    right = np.ascontiguousarray(right, dtype=np.float64)
    out = np.zeros(len(right))
    ok = load().orc_gauss_solve(_dptr(left), _dptr(right), len(right), _dptr(out))
    return out if ok else None


def kde(scores, decoys, monotonic=True, bins=1000, bw_mult=1.0, queries=(), det=False):
    """kde::Builder::build + Estimator::posterior_error -> (bins, min_score, score_step, pep(queries))"""
    scores = np.ascontiguousarray(scores, dtype=np.float64)
    decoys = np.ascontiguousarray(decoys, dtype=np.uint8)
    q = np.ascontiguousarray(queries, dtype=np.float64)
    out_bins, ms, pep = np.zeros(bins), np.zeros(2), np.zeros(len(q))
    with _mode(det):
        load().orc_kde(_dptr(scores), L.as_ptr(decoys, C.c_uint8), len(scores), int(monotonic), bins, bw_mult, _dptr(out_bins),
                       _dptr(ms), _dptr(q), len(q), _dptr(pep))
    return out_bins, ms[0], ms[1], pep


def rescore(features, precursor_tol, peptide_key, n_peptide_keys, protein_key, n_protein_keys, aligned_rt=None,
            delta_rt_model=None, delta_ims_model=None, want_rows=False, det=True):
    """spectrum_fdr + picked_peptide + picked_protein (runner.rs:536-541); dict of arrays in input order.
    det=True (default): the arithmetic contract the device evaluates; det=False: the reference's own order + platform libm."""
    f = np.ascontiguousarray(features, dtype=L.FEATURE_DTYPE).reshape(-1)
    n = len(f)
    pk = np.ascontiguousarray(peptide_key, dtype=np.uint32)
    prk = np.ascontiguousarray(protein_key, dtype=np.uint32)
    opt = [None if a is None else np.ascontiguousarray(a, dtype=np.float32) for a in (aligned_rt, delta_rt_model, delta_ims_model)]
    outs = {k: np.empty(n, np.float32) for k in ("discriminant_score", "posterior_error", "spectrum_q", "peptide_q", "protein_q")}
    order = np.empty(n, np.uint32)
    passing = np.zeros(3, np.uint64)
    coef = np.zeros(20)
    rows = np.zeros((n, 20)) if want_rows else None
    t = precursor_tol.to_c()
    with _mode(det):
        fitted = load().orc_rescore(f.ctypes.data, n, t.kind, t.lo, t.hi, *[None if a is None else L.as_ptr(a, C.c_float) for a in opt],
                                    L.as_ptr(pk, C.c_uint32), n_peptide_keys, L.as_ptr(prk, C.c_uint32), n_protein_keys,
                                    *[L.as_ptr(outs[k], C.c_float) for k in ("discriminant_score", "posterior_error", "spectrum_q",
                                                                               "peptide_q", "protein_q")],
                                    L.as_ptr(order, C.c_uint32), L.as_ptr(passing, C.c_uint64), _dptr(coef),
                                    None if rows is None else _dptr(rows))
    return dict(outs, order=order, passing=passing, coef=coef, lda_fitted=bool(fitted), rows=rows)


def fasta_num_targets(fasta_text, params):
    p, keep = params.to_c()
    return int(load().orc_fasta_num_targets(fasta_text.encode(), C.byref(p)))


def prefilter_chunk_size(fasta_text, params, requested=0):
    p, keep = params.to_c()
    return int(load().orc_prefilter_chunk_size(fasta_text.encode(), C.byref(p), requested))


# ---- the predict_rt block (runner.rs:513-530) ------------------------------------------------------------------------------
def linreg_fit(rows, y, keep=None):
    """LinearRegression::fit (regression.rs:58-122) -> (beta, r2) or None"""
    rows = np.ascontiguousarray(rows, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    keep = np.ones(len(y), np.uint8) if keep is None else np.ascontiguousarray(keep, dtype=np.uint8)
    beta, r2 = np.zeros(rows.shape[1]), np.zeros(1)
    ok = load().orc_linreg_fit(_dptr(rows), _dptr(y), L.as_ptr(keep, C.c_uint8), len(y), rows.shape[1], _dptr(beta), _dptr(r2))
    return (beta, float(r2[0])) if ok else None


def rt_embed(sequence: str, mono: float):
    seq = np.frombuffer(sequence.encode(), dtype=np.uint8).copy()
    out = np.zeros(69)
    load().orc_rt_embed(L.as_ptr(seq, C.c_uint8), len(seq), mono, _dptr(out))
    return out


def im_embed(sequence: str, mono: float, charge: int):
    seq = np.frombuffer(sequence.encode(), dtype=np.uint8).copy()
    out = np.zeros(100)
    load().orc_im_embed(L.as_ptr(seq, C.c_uint8), len(seq), mono, charge, _dptr(out))
    return out


def predict_rt(features, n_files, seq_off, seq, mono):
    """runner.rs:513-530: poisson-sorted q-values, global_alignment, retention / mobility models; dict of arrays, input order"""
    f = np.ascontiguousarray(features, dtype=L.FEATURE_DTYPE).reshape(-1)
    n = len(f)
    seq_off = np.ascontiguousarray(seq_off, dtype=np.uint64)
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    mono = np.ascontiguousarray(mono, dtype=np.float32)
    names = ("spectrum_q", "aligned_rt", "predicted_rt", "delta_rt_model", "predicted_ims", "delta_ims_model")
    outs = {k: np.empty(n, np.float32) for k in names}
    align = np.zeros((n_files, 3), np.float32)
    fitted = np.zeros(2, np.int32)
    r2 = np.zeros(2)
    load().orc_predict_rt(f.ctypes.data, n, n_files, L.as_ptr(seq_off, C.c_uint64), L.as_ptr(seq, C.c_uint8),
                          L.as_ptr(mono, C.c_float), *[L.as_ptr(outs[k], C.c_float) for k in names], L.as_ptr(align, C.c_float),
                          fitted.ctypes.data_as(C.POINTER(C.c_int32)), _dptr(r2))
    return dict(outs, alignments=align, fitted=fitted.astype(bool), r2=r2)


def log_mode(kind=None):
    """0: the oracle's ln() is the platform libm's; 1: correctly rounded (libquadmath) — the default of the checker build."""
    return load(kind).orc_get_log_mode()


class LogMode:
    """with oracle_lib.LogMode(0): ...  — the oracle with the platform libm's ln() (the reference's arithmetic on this host)."""

    def __init__(self, mode, kind=None):
        self.mode, self.kind = mode, kind

    def __enter__(self):
        self.prev = log_mode(self.kind)
        load(self.kind).orc_set_log_mode(self.mode)
        return self

    def __exit__(self, *a):
        load(self.kind).orc_set_log_mode(self.prev)


def ln_batch(x, mode):
    import numpy as np
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    dp = C.POINTER(C.c_double)
    load().orc_ln_batch(mode, x.ctypes.data_as(dp), len(x), out.ctypes.data_as(dp))
    return out
