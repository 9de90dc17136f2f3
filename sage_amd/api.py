"""Host-side mirror of the sage-core API for the search-and-score path, over the C ABI.

Names and argument meaning follow the reference (crates/sage/src):
  DatabaseParameters  ~ database.rs:59-139   Builder / Parameters (JSON `database` section)
  IndexedDatabase     ~ database.rs:384-395
  SpectrumProcessor   ~ spectrum.rs:263-413
  Scorer              ~ scoring.rs:210-309
All compute happens in libsage_hip.so; nothing here falls back to Python arithmetic.
"""
import ctypes as C
import weakref
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib as L


@dataclass
class Tolerance:  # mass.rs:10-16
    kind: str  # "ppm" | "pct" | "da"
    lo: float
    hi: float

    @staticmethod
    def from_json(obj) -> "Tolerance":
        (k, v), = obj.items()
        return Tolerance(k, float(v[0]), float(v[1]))

    def to_c(self):
        return L.SageTolerance(L.TOL_KINDS[self.kind], self.lo, self.hi)


@dataclass
class DatabaseParameters:
    """database.rs:59-93 Builder: None == field absent from the JSON."""
    bucket_size: Optional[int] = None
    enzyme: Optional[dict] = None  # keys: missed_cleavages,min_len,max_len,cleave_at,restrict,c_terminal,semi_enzymatic
    peptide_min_mass: Optional[float] = None
    peptide_max_mass: Optional[float] = None
    ion_kinds: Optional[List[str]] = None
    min_ion_index: Optional[int] = None
    static_mods: Optional[Dict[str, float]] = None
    variable_mods: Optional[Dict[str, List[float]]] = None
    max_variable_mods: Optional[int] = None
    decoy_tag: Optional[str] = None
    generate_decoys: Optional[bool] = None
    fasta: Optional[str] = None
    prefilter: Optional[bool] = None             # database.rs:88-92: search the FASTA in chunks first (runner.rs:104-127)
    prefilter_chunk_size: Optional[int] = None   # target proteins per chunk; None / 0 = auto (database.rs:142-160)
    prefilter_low_memory: Optional[bool] = None  # default true (database.rs:113)
    peptides_only: bool = False  # ours: stop after reorder_peptides; the fragment index is then built on the device

    @staticmethod
    def from_json(obj: dict) -> "DatabaseParameters":
        known = DatabaseParameters.__dataclass_fields__.keys()
        return DatabaseParameters(**{k: v for k, v in obj.items() if k in known})  # unknown keys ignored (serde)

    def to_c(self, struct_type=L.SageDbParams):
        """Builder::make_parameters defaults (database.rs:96-115).  Returns (struct, keepalive)."""
        keep = []
        p = struct_type()
        p.bucket_size = self.bucket_size if self.bucket_size is not None else 8192
        e = self.enzyme
        p.enzyme_present = 0 if e is None else 1
        e = e or {}

        def opt_int(key):
            v = e.get(key)
            return -1 if v is None else int(v)

        def opt_str(key):
            v = e.get(key)
            if v is None:
                return None
            b = v.encode()
            keep.append(b)
            return b

        p.missed_cleavages = opt_int("missed_cleavages")
        p.min_len = opt_int("min_len")
        p.max_len = opt_int("max_len")
        p.cleave_at = opt_str("cleave_at")
        p.restrict_ = opt_str("restrict")
        p.c_terminal = opt_int("c_terminal")
        p.semi_enzymatic = opt_int("semi_enzymatic")
        p.peptide_min_mass = 500.0 if self.peptide_min_mass is None else self.peptide_min_mass
        p.peptide_max_mass = 5000.0 if self.peptide_max_mass is None else self.peptide_max_mass
        kinds = np.array([L.ION_KINDS[k] for k in (self.ion_kinds if self.ion_kinds is not None else ["b", "y"])],
                         dtype=np.uint8)
        keep.append(kinds)
        p.ion_kinds = L.as_ptr(kinds, C.c_uint8)
        p.n_ion_kinds = len(kinds)
        p.min_ion_index = 2 if self.min_ion_index is None else self.min_ion_index
        sm = list((self.static_mods or {}).items())
        sk = (C.c_char_p * max(len(sm), 1))(*[k.encode() for k, _ in sm])
        sv = np.array([v for _, v in sm], dtype=np.float32)
        keep += [sk, sv]
        p.static_mod_keys = sk
        p.static_mod_masses = L.as_ptr(sv, C.c_float)
        p.n_static_mods = len(sm)
        vm = [(k, m) for k, ms in (self.variable_mods or {}).items()
              for m in (ms if isinstance(ms, (list, tuple)) else [ms])]
        vk = (C.c_char_p * max(len(vm), 1))(*[k.encode() for k, _ in vm])
        vv = np.array([m for _, m in vm], dtype=np.float32)
        keep += [vk, vv]
        p.var_mod_keys = vk
        p.var_mod_masses = L.as_ptr(vv, C.c_float)
        p.n_var_mods = len(vm)
        p.max_variable_mods = 2 if self.max_variable_mods is None else max(int(self.max_variable_mods), 1)
        tag = (self.decoy_tag if self.decoy_tag is not None else "rev_").encode()
        keep.append(tag)
        p.decoy_tag = tag
        p.generate_decoys = 1 if (self.generate_decoys is None or self.generate_decoys) else 0
        p.peptides_only = int(self.peptides_only)
        return p, keep

    def build(self, fasta_text: str, peptides_only: Optional[bool] = None) -> "IndexedDatabase":
        """Parameters::build(Fasta::parse(..)) — database.rs:260-263.  peptides_only: digest / modify / sort / dedup on the
        host and leave build_from_peptides (fragments, sort) to DeviceDatabase (index_build.hip)."""
        lib = L.load()
        p, keep = self.to_c()
        if peptides_only is not None:
            p.peptides_only = int(peptides_only)
        h = C.c_void_p()
        L.check(lib.sage_hip_hostdb_build(fasta_text.encode(), C.byref(p), C.byref(h)))
        return IndexedDatabase(h)

    # ---- the `prefilter` flow of sage-cli (runner.rs:104-127, :143-238) ----
    def num_targets(self, fasta_text: str) -> int:
        """Fasta::parse(..).targets.len()"""
        p, keep = self.to_c()
        n = C.c_uint64()
        L.check(L.load().sage_hip_fasta_num_targets(fasta_text.encode(), C.byref(p), C.byref(n)))
        return int(n.value)

    def auto_prefilter_chunk_size(self, fasta_text: str) -> int:
        """Parameters::auto_calculate_prefilter_chunk_size (database.rs:142-160)"""
        p, keep = self.to_c()
        n = C.c_uint64()
        L.check(L.load().sage_hip_prefilter_chunk_size(fasta_text.encode(), C.byref(p), int(self.prefilter_chunk_size or 0),
                                                       C.byref(n)))
        return int(n.value)

    def build_chunk(self, fasta_text: str, first_target: int, n_targets: int, peptides_only: Optional[bool] = None):
        """Parameters::build over one chunk of Fasta::iter_chunks (fasta.rs:81-89)"""
        p, keep = self.to_c()
        if peptides_only is not None:
            p.peptides_only = int(peptides_only)
        h = C.c_void_p()
        L.check(L.load().sage_hip_hostdb_build_chunk(fasta_text.encode(), C.byref(p), first_target, n_targets, C.byref(h)))
        return IndexedDatabase(h)

    def merge_kept(self, chunks, keeps, peptides_only: Optional[bool] = None) -> "IndexedDatabase":
        """runner.rs:215-238: the peptides quick_score kept in every chunk database -> reorder_peptides -> build_from_peptides"""
        p, keep = self.to_c()
        if peptides_only is not None:
            p.peptides_only = int(peptides_only)
        masks = [np.ascontiguousarray(k, dtype=np.uint8) for k in keeps]
        assert all(len(m) == c.n_peptides for m, c in zip(masks, chunks))
        hs = (C.c_void_p * max(len(chunks), 1))(*[c._h for c in chunks])
        ms = (L.c_u8_p * max(len(chunks), 1))(*[L.as_ptr(m, C.c_uint8) for m in masks])
        h = C.c_void_p()
        L.check(L.load().sage_hip_hostdb_merge_kept(hs, ms, len(chunks), C.byref(p), C.byref(h)))
        return IndexedDatabase(h)


def _view_array(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(C.addressof(ptr.contents))
    return np.frombuffer(buf, dtype=dtype, count=n)


class IndexedDatabase:
    """database.rs:384-395 — flat views over the host index built by libsage_hip."""

    def __init__(self, handle):
        self._h = handle
        lib = L.load()
        v = L.SageDbView()
        L.check(lib.sage_hip_hostdb_view(self._h, C.byref(v)))
        self._view = v
        self.n_peptides = int(v.n_peptides)
        self.n_fragments = int(v.n_fragments)
        self.bucket_size = int(v.bucket_size)
        self.has_fragments = bool(v.fragments)
        self.fragments = _view_array(v.fragments, self.n_fragments, L.THEORETICAL_DTYPE) if self.has_fragments else \
            np.zeros(0, dtype=L.THEORETICAL_DTYPE)
        self.min_value = _view_array(v.min_value, int(v.n_buckets), np.float32)
        self.pep_mono = _view_array(v.pep_mono, self.n_peptides, np.float32)
        self.seq_off = _view_array(v.seq_off, self.n_peptides + 1, np.uint64)
        total = int(self.seq_off[-1]) if self.n_peptides else 0
        self.seq = _view_array(v.seq, total, np.uint8)
        self.mods = _view_array(v.mods, total, np.float32)
        self.nterm = _view_array(v.nterm, self.n_peptides, np.float32)
        self.cterm = _view_array(v.cterm, self.n_peptides, np.float32)
        self.decoy = _view_array(v.decoy, self.n_peptides, np.uint8)
        self.missed_cleavages = _view_array(v.missed_cleavages, self.n_peptides, np.uint8)
        self.ion_kinds = _view_array(v.ion_kinds, int(v.n_ion_kinds), np.uint8)

    def size(self):  # database.rs:427-429
        return self.n_fragments

    def buckets(self):  # database.rs:431-433
        return self.min_value

    def peptide_string(self, i: int) -> str:
        lib = L.load()
        n = lib.sage_hip_hostdb_peptide_string(self._h, i, None, 0)
        buf = C.create_string_buffer(int(n))
        lib.sage_hip_hostdb_peptide_string(self._h, i, buf, n)
        return buf.value.decode()

    def peptide_proteins(self, i: int) -> str:
        lib = L.load()
        n = lib.sage_hip_hostdb_peptide_proteins(self._h, i, None, 0)
        buf = C.create_string_buffer(int(n))
        lib.sage_hip_hostdb_peptide_proteins(self._h, i, buf, n)
        return buf.value.decode()

    def peptide_info(self, i: int):
        """(Peptide.proteins.len(), Peptide.semi_enzymatic)"""
        n, semi = C.c_uint32(), C.c_uint8()
        L.check(L.load().sage_hip_hostdb_peptide_info(self._h, i, C.byref(n), C.byref(semi)))
        return int(n.value), int(semi.value)

    def competition_keys(self, peptide_idx):
        """Map keys of fdr::picked_peptide / picked_protein (fdr.rs:126-132, :158-161) for PSMs of the given peptides:
        (peptide_key[n], n_peptide_keys, protein_key[n], n_protein_keys), dense ids, 0xFFFFFFFF = shared peptide."""
        idx = np.ascontiguousarray(peptide_idx, dtype=np.uint32)
        pk, prk = np.empty(len(idx), np.uint32), np.empty(len(idx), np.uint32)
        npk, nprk = C.c_uint32(), C.c_uint32()
        L.check(L.load().sage_hip_hostdb_competition_keys(self._h, L.as_ptr(idx, C.c_uint32), len(idx), L.as_ptr(pk, C.c_uint32),
                                                          C.byref(npk), L.as_ptr(prk, C.c_uint32), C.byref(nprk)))
        return pk, int(npk.value), prk, int(nprk.value)

    def feature_peptides(self, peptide_idx):
        """(seq_off[n + 1], residues, monoisotopic[n]) of the peptides behind `n` PSMs — the per-Feature peptide data the
        retention-time / mobility models embed (retention_model.rs:44-62, mobility_model.rs:103-158)."""
        idx = np.ascontiguousarray(peptide_idx, dtype=np.uint32)
        off = np.zeros(len(idx) + 1, dtype=np.uint64)
        lib = L.load()
        L.check(lib.sage_hip_hostdb_feature_peptides(self._h, L.as_ptr(idx, C.c_uint32), len(idx), L.as_ptr(off, C.c_uint64), None, None))
        seq = np.zeros(max(int(off[-1]), 1), dtype=np.uint8)
        mono = np.zeros(len(idx), dtype=np.float32)
        L.check(lib.sage_hip_hostdb_feature_peptides(self._h, L.as_ptr(idx, C.c_uint32), len(idx), L.as_ptr(off, C.c_uint64),
                                                     L.as_ptr(seq, C.c_uint8), L.as_ptr(mono, C.c_float)))
        return off, seq, mono

    def sequence(self, i: int) -> str:
        return bytes(self.seq[int(self.seq_off[i]):int(self.seq_off[i + 1])]).decode()

    def to_device(self, device: int = 0) -> "DeviceDatabase":
        return DeviceDatabase(self, device)

    def close(self):
        if self._h:
            L.load().sage_hip_hostdb_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceDatabase:
    def __init__(self, host: IndexedDatabase, device: int = 0, build_on_device: bool = False):
        """build_on_device: ignore the host's fragments (if any) and generate the index from the peptide list on the GPU
        (Parameters::build_from_peptides, database.rs:265-346); implied when the host database is peptides-only."""
        lib = L.load()
        self.host = host
        self.device = device
        self._h = C.c_void_p()
        view = host._view
        if build_on_device and host.has_fragments:
            view = L.SageDbView()
            C.memmove(C.byref(view), C.byref(host._view), C.sizeof(L.SageDbView))
            view.fragments = None
            view.n_fragments = 0
        L.check(lib.sage_hip_db_create(C.byref(view), device, C.byref(self._h)))
        self.device_bytes = int(lib.sage_hip_db_device_bytes(self._h))

    def close(self):
        if self._h:
            L.load().sage_hip_db_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class RawSpectrum:  # spectrum.rs:81-106 (MS2, centroided)
    mz: np.ndarray
    intensity: np.ndarray
    precursor_mz: float
    precursor_charge: Optional[int] = None
    isolation_window: Optional[Tuple[float, float]] = None  # Tolerance::Da(lo, hi)
    scan_start_time: float = 0.0
    inverse_ion_mobility: Optional[float] = None
    file_id: int = 0
    id: str = ""


@dataclass
class ProcessedSpectrum:  # spectrum.rs:57-79
    masses: np.ndarray
    intensities: np.ndarray
    total_ion_current: float
    precursor_mz: float
    precursor_charge: Optional[int] = None
    isolation_window: Optional[Tuple[float, float]] = None
    scan_start_time: float = 0.0
    inverse_ion_mobility: Optional[float] = None
    file_id: int = 0
    id: str = ""


class SpectrumProcessor:  # spectrum.rs:263-413
    def __init__(self, take_top_n: int, deisotope: bool, min_deisotope_mz: float = 0.0):
        self.take_top_n = take_top_n
        self.deisotope = deisotope
        self.min_deisotope_mz = min_deisotope_mz

    def process(self, raw: RawSpectrum) -> ProcessedSpectrum:
        lib = L.load()
        mz = np.ascontiguousarray(raw.mz, dtype=np.float32)
        it = np.ascontiguousarray(raw.intensity, dtype=np.float32)
        n = len(mz)
        om = np.empty(max(n, 1), dtype=np.float32)
        oi = np.empty(max(n, 1), dtype=np.float32)
        tic = C.c_float()
        k = lib.sage_hip_process_ms2(self.take_top_n, int(self.deisotope), self.min_deisotope_mz,
                                     L.as_ptr(mz, C.c_float), L.as_ptr(it, C.c_float), n,
                                     raw.precursor_charge or 0, L.as_ptr(om, C.c_float), L.as_ptr(oi, C.c_float),
                                     C.byref(tic))
        k = int(k)
        return ProcessedSpectrum(om[:k].copy(), oi[:k].copy(), float(np.float32(tic.value)), raw.precursor_mz,
                                 raw.precursor_charge, raw.isolation_window, raw.scan_start_time,
                                 raw.inverse_ion_mobility, raw.file_id, raw.id)


class SpectrumBatch:
    """SoA batch of ProcessedSpectrum (SageSpectrumBatch)."""

    def __init__(self, peak_off, masses, intensities, precursor_mz, precursor_charge, total_ion_current,
                 isolation_lo=None, isolation_hi=None, scan_start_time=None, inverse_ion_mobility=None, file_id=None):
        self.peak_off = np.ascontiguousarray(peak_off, dtype=np.uint64)
        self.masses = np.ascontiguousarray(masses, dtype=np.float32)
        self.intensities = np.ascontiguousarray(intensities, dtype=np.float32)
        self.precursor_mz = np.ascontiguousarray(precursor_mz, dtype=np.float32)
        self.precursor_charge = np.ascontiguousarray(precursor_charge, dtype=np.uint8)
        self.total_ion_current = np.ascontiguousarray(total_ion_current, dtype=np.float32)
        f32 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float32)
        self.isolation_lo, self.isolation_hi = f32(isolation_lo), f32(isolation_hi)
        self.scan_start_time, self.inverse_ion_mobility = f32(scan_start_time), f32(inverse_ion_mobility)
        self.file_id = None if file_id is None else np.ascontiguousarray(file_id, dtype=np.uint32)
        self.n = len(self.precursor_mz)
        assert len(self.peak_off) == self.n + 1

    @staticmethod
    def from_spectra(spectra: Sequence[ProcessedSpectrum]) -> "SpectrumBatch":
        n = len(spectra)
        off = np.zeros(n + 1, dtype=np.uint64)
        for i, s in enumerate(spectra):
            off[i + 1] = off[i] + len(s.masses)
        cat = lambda xs: np.concatenate(xs).astype(np.float32) if n else np.zeros(0, np.float32)
        nan = float("nan")
        iso = [s.isolation_window for s in spectra]
        return SpectrumBatch(
            off, cat([s.masses for s in spectra]), cat([s.intensities for s in spectra]),
            [s.precursor_mz for s in spectra], [s.precursor_charge or 0 for s in spectra],
            [s.total_ion_current for s in spectra],
            [w[0] if w else nan for w in iso], [w[1] if w else nan for w in iso],
            [s.scan_start_time for s in spectra],
            [nan if s.inverse_ion_mobility is None else s.inverse_ion_mobility for s in spectra],
            [s.file_id for s in spectra])

    def page_locked(self) -> "SpectrumBatch":
        """The same batch with every array in page-locked host memory (sage_hip_host_alloc): Scorer.score then moves the
        peaks by DMA straight out of these arrays instead of staging them.  What a caller that owns its spectrum arena
        (the mzML reader, a Rust Vec allocated through the C ABI) would hand over."""
        lib = L.load()

        def lock(a):
            if a is None:
                return None
            # the block lives exactly as long as something refers to the array made over it (a view, a to_c() struct holding
            # the array, this batch): the ctypes buffer is the ndarray's base, and its finalizer frees the block
            p = C.c_void_p()
            L.check(lib.sage_hip_host_alloc(max(a.nbytes, 1), C.byref(p)))
            buf = (C.c_char * max(a.nbytes, 1)).from_address(p.value)
            weakref.finalize(buf, lib.sage_hip_host_free, C.c_void_p(p.value))
            out = np.frombuffer(buf, dtype=a.dtype, count=a.size).reshape(a.shape)
            out[...] = a
            return out

        b = SpectrumBatch.__new__(SpectrumBatch)
        for k in ("peak_off", "masses", "intensities", "precursor_mz", "precursor_charge", "total_ion_current", "isolation_lo",
                  "isolation_hi", "scan_start_time", "inverse_ion_mobility", "file_id"):
            setattr(b, k, lock(getattr(self, k)))
        b.n = self.n
        return b

    def subset(self, idx) -> "SpectrumBatch":
        idx = np.asarray(idx)
        lens = (self.peak_off[1:] - self.peak_off[:-1])[idx].astype(np.int64)
        off = np.zeros(len(idx) + 1, dtype=np.uint64)
        off[1:] = np.cumsum(lens)
        starts = self.peak_off[:-1][idx].astype(np.int64)
        gather = (np.repeat(starts - off[:-1].astype(np.int64), lens) + np.arange(int(off[-1]))) if len(idx) else np.zeros(0, np.int64)
        pick = lambda a: None if a is None else a[idx]
        return SpectrumBatch(off, self.masses[gather], self.intensities[gather], self.precursor_mz[idx],
                             self.precursor_charge[idx], self.total_ion_current[idx], pick(self.isolation_lo),
                             pick(self.isolation_hi), pick(self.scan_start_time), pick(self.inverse_ion_mobility),
                             pick(self.file_id))

    def to_c(self, struct_type=L.SageSpectrumBatch):
        b = struct_type()
        b.n_spectra = self.n
        p = lambda a, t: None if a is None else L.as_ptr(a, t)
        b.peak_off = p(self.peak_off, C.c_uint64)
        b.masses = p(self.masses, C.c_float)
        b.intensities = p(self.intensities, C.c_float)
        b.precursor_mz = p(self.precursor_mz, C.c_float)
        b.precursor_charge = p(self.precursor_charge, C.c_uint8)
        b.isolation_lo = p(self.isolation_lo, C.c_float)
        b.isolation_hi = p(self.isolation_hi, C.c_float)
        b.total_ion_current = p(self.total_ion_current, C.c_float)
        b.scan_start_time = p(self.scan_start_time, C.c_float)
        b.inverse_ion_mobility = p(self.inverse_ion_mobility, C.c_float)
        b.file_id = p(self.file_id, C.c_uint32)
        return b


@dataclass
class ScorerParams:
    """scoring.rs:210-232 Scorer fields (minus db); defaults = sage-cli input.rs:355-385."""
    precursor_tol: Tolerance = field(default_factory=lambda: Tolerance("ppm", -10.0, 10.0))
    fragment_tol: Tolerance = field(default_factory=lambda: Tolerance("ppm", -10.0, 10.0))
    min_matched_peaks: int = 4
    min_isotope_err: int = 0
    max_isotope_err: int = 0
    min_precursor_charge: int = 2
    max_precursor_charge: int = 4
    override_precursor_charge: bool = False
    max_fragment_charge: Optional[int] = None
    chimera: bool = False
    report_psms: int = 1
    wide_window: bool = False
    annotate_matches: bool = False
    score_type: str = "SageHyperScore"

    def to_c(self, struct_type=L.SageScorerParams):
        p = struct_type()
        p.precursor_tol = self.precursor_tol.to_c()
        p.fragment_tol = self.fragment_tol.to_c()
        p.min_matched_peaks = self.min_matched_peaks
        p.min_isotope_err = self.min_isotope_err
        p.max_isotope_err = self.max_isotope_err
        p.min_precursor_charge = self.min_precursor_charge
        p.max_precursor_charge = self.max_precursor_charge
        p.override_precursor_charge = int(self.override_precursor_charge)
        p.chimera = int(self.chimera)
        p.max_fragment_charge = -1 if self.max_fragment_charge is None else self.max_fragment_charge
        p.wide_window = int(self.wide_window)
        p.annotate_matches = int(self.annotate_matches)
        p.report_psms = self.report_psms
        p.score_type = L.SCORE_TYPES[self.score_type]
        return p


class RawBatch:
    """SoA batch of RawSpectrum (SageRawBatch): centroided MS2 peaks + precursors[0]."""

    def __init__(self, spectra: Sequence[RawSpectrum]):
        n = len(spectra)
        self.n = n
        self.ids = [s.id for s in spectra]
        self.peak_off = np.zeros(n + 1, dtype=np.uint64)
        for i, s in enumerate(spectra):
            self.peak_off[i + 1] = self.peak_off[i] + len(s.mz)
        cat = lambda xs: np.ascontiguousarray(np.concatenate(xs), dtype=np.float32) if n else np.zeros(0, np.float32)
        self.mz = cat([np.asarray(s.mz, np.float32) for s in spectra])
        self.intensities = cat([np.asarray(s.intensity, np.float32) for s in spectra])
        f32 = lambda xs: np.ascontiguousarray(xs, dtype=np.float32)
        nan = float("nan")
        self.precursor_mz = f32([s.precursor_mz for s in spectra])
        self.precursor_charge = np.ascontiguousarray([s.precursor_charge or 0 for s in spectra], dtype=np.uint8)
        self.isolation_lo = f32([s.isolation_window[0] if s.isolation_window else nan for s in spectra])
        self.isolation_hi = f32([s.isolation_window[1] if s.isolation_window else nan for s in spectra])
        self.scan_start_time = f32([s.scan_start_time for s in spectra])
        self.inverse_ion_mobility = f32([nan if s.inverse_ion_mobility is None else s.inverse_ion_mobility for s in spectra])
        self.file_id = np.ascontiguousarray([s.file_id for s in spectra], dtype=np.uint32)

    @classmethod
    def from_arrays(cls, ids, peak_off, mz, intensities, precursor_mz, precursor_charge, isolation_lo, isolation_hi,
                    scan_start_time, inverse_ion_mobility, file_id) -> "RawBatch":
        """Adopt SoA arrays (NaN == None for the optional floats, 0 == unknown charge): what the C++ mzML reader returns."""
        b = cls.__new__(cls)
        b.n = len(precursor_mz)
        b.ids = list(ids)
        b.peak_off = np.ascontiguousarray(peak_off, dtype=np.uint64)
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        b.mz, b.intensities = f32(mz), f32(intensities)
        b.precursor_mz = f32(precursor_mz)
        b.precursor_charge = np.ascontiguousarray(precursor_charge, dtype=np.uint8)
        b.isolation_lo, b.isolation_hi = f32(isolation_lo), f32(isolation_hi)
        b.scan_start_time, b.inverse_ion_mobility = f32(scan_start_time), f32(inverse_ion_mobility)
        b.file_id = np.ascontiguousarray(file_id, dtype=np.uint32)
        assert len(b.peak_off) == b.n + 1 and len(b.mz) == len(b.intensities) == int(b.peak_off[-1])
        return b

    def slice(self, begin: int, end: int) -> "RawBatch":
        """Spectra [begin, end) as a batch of their own (contiguous shard of a file)."""
        a, b = int(self.peak_off[begin]), int(self.peak_off[end])
        return RawBatch.from_arrays(self.ids[begin:end], self.peak_off[begin:end + 1] - np.uint64(a), self.mz[a:b], self.intensities[a:b],
                                    self.precursor_mz[begin:end], self.precursor_charge[begin:end], self.isolation_lo[begin:end],
                                    self.isolation_hi[begin:end], self.scan_start_time[begin:end],
                                    self.inverse_ion_mobility[begin:end], self.file_id[begin:end])

    def subset(self, idx) -> "RawBatch":
        """The spectra at positions `idx` as a batch of their own (a shard that is not contiguous in the file: sharding.plan_mass_shards)."""
        idx = np.asarray(idx, dtype=np.int64)
        lens = (self.peak_off[1:] - self.peak_off[:-1])[idx].astype(np.int64)
        off = np.zeros(len(idx) + 1, dtype=np.uint64)
        off[1:] = np.cumsum(lens)
        starts = self.peak_off[:-1][idx].astype(np.int64)
        gather = (np.repeat(starts - off[:-1].astype(np.int64), lens) + np.arange(int(off[-1]))) if len(idx) else np.zeros(0, np.int64)
        return RawBatch.from_arrays([self.ids[i] for i in idx], off, self.mz[gather], self.intensities[gather], self.precursor_mz[idx],
                                    self.precursor_charge[idx], self.isolation_lo[idx], self.isolation_hi[idx], self.scan_start_time[idx],
                                    self.inverse_ion_mobility[idx], self.file_id[idx])

    def spectrum(self, i: int) -> RawSpectrum:
        lo, hi = int(self.peak_off[i]), int(self.peak_off[i + 1])
        iso = None if np.isnan(self.isolation_lo[i]) else (float(self.isolation_lo[i]), float(self.isolation_hi[i]))
        ims = None if np.isnan(self.inverse_ion_mobility[i]) else float(self.inverse_ion_mobility[i])
        return RawSpectrum(self.mz[lo:hi], self.intensities[lo:hi], float(self.precursor_mz[i]),
                           int(self.precursor_charge[i]) or None, iso, float(self.scan_start_time[i]), ims, int(self.file_id[i]),
                           self.ids[i])

    def to_c(self):
        b = L.SageRawBatch()
        b.n_spectra = self.n
        b.peak_off = L.as_ptr(self.peak_off, C.c_uint64)
        b.mz = L.as_ptr(self.mz, C.c_float)
        b.intensities = L.as_ptr(self.intensities, C.c_float)
        b.precursor_mz = L.as_ptr(self.precursor_mz, C.c_float)
        b.precursor_charge = L.as_ptr(self.precursor_charge, C.c_uint8)
        b.isolation_lo = L.as_ptr(self.isolation_lo, C.c_float)
        b.isolation_hi = L.as_ptr(self.isolation_hi, C.c_float)
        b.scan_start_time = L.as_ptr(self.scan_start_time, C.c_float)
        b.inverse_ion_mobility = L.as_ptr(self.inverse_ion_mobility, C.c_float)
        b.file_id = L.as_ptr(self.file_id, C.c_uint32)
        return b


class DeviceBatch:
    def __init__(self, scorer: "Scorer", batch: Optional[SpectrumBatch], handle=None, n: int = 0):
        lib = L.load()
        self._keep = batch
        if handle is not None:  # adopted (Scorer.process_upload)
            self._h, self.n = handle, n
            return
        self.n = batch.n
        self._h = C.c_void_p()
        cb = batch.to_c()
        L.check(lib.sage_hip_batch_upload(scorer._h, C.byref(cb), C.byref(self._h)))

    def download(self):
        """(peak_off[n+1], masses, intensities, total_ion_current[n]) of the resident ProcessedSpectrum arrays."""
        lib = L.load()
        off = np.zeros(self.n + 1, dtype=np.uint64)
        L.check(lib.sage_hip_batch_download(self._h, L.as_ptr(off, C.c_uint64), None, None, None))
        total = int(off[-1])
        m, it, tic = np.zeros(max(total, 1), np.float32), np.zeros(max(total, 1), np.float32), np.zeros(max(self.n, 1), np.float32)
        L.check(lib.sage_hip_batch_download(self._h, L.as_ptr(off, C.c_uint64), L.as_ptr(m, C.c_float), L.as_ptr(it, C.c_float),
                                            L.as_ptr(tic, C.c_float)))
        return off, m[:total], it[:total], tic[:self.n]

    def close(self):
        if self._h:
            L.load().sage_hip_batch_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Scorer:
    """scoring.rs:210-309.  `score` takes a whole batch: a per-spectrum call cannot amortise a launch."""

    def __init__(self, db: DeviceDatabase, params: ScorerParams):
        lib = L.load()
        self.db = db
        self.params = params
        self._h = C.c_void_p()
        cp = params.to_c()
        L.check(lib.sage_hip_scorer_create(db._h, C.byref(cp), C.byref(self._h)))

    def clone(self) -> "Scorer":
        """A second handle on the same device database (own streams and working set): `&Scorer` is shared by every rayon
        worker in the reference (scoring.rs:300); calls on one handle serialise, clones run concurrently."""
        other = Scorer.__new__(Scorer)
        other.db, other.params, other._h = self.db, self.params, C.c_void_p()
        L.check(L.load().sage_hip_scorer_clone(self._h, C.byref(other._h)))
        return other

    def upload(self, batch: SpectrumBatch) -> DeviceBatch:
        return DeviceBatch(self, batch)

    def process_upload(self, raw: "RawBatch", take_top_n: int = 150, deisotope: bool = True, min_deisotope_mz: float = 0.0,
                       min_peaks: int = 15):
        """SpectrumProcessor::process (spectrum.rs:279-412) + the min_peaks filter of runner.rs:313 on the device; the
        processed batch stays resident.  Returns (DeviceBatch, peaks kept per spectrum before the filter)."""
        lib = L.load()
        h = C.c_void_p()
        npk = np.zeros(max(raw.n, 1), dtype=np.uint32)
        cb = raw.to_c()
        L.check(lib.sage_hip_batch_process_upload(self._h, C.byref(cb), take_top_n, int(deisotope), min_deisotope_mz, min_peaks,
                                                  C.byref(h), L.as_ptr(npk, C.c_uint32)))
        return DeviceBatch(self, None, handle=h, n=raw.n), npk[:raw.n]

    def _alloc_out(self, n):
        """Output arrays in page-locked host memory (sage_hip_host_alloc), reused while the size is unchanged.
        NOTE: a later call with the same batch size overwrites the arrays returned by the previous one."""
        key = (n, self.params.report_psms)
        cached = getattr(self, "_pinned", None)
        if cached is None or cached[0] != key:
            self._free_pinned()
            lib = L.load()
            nb_f = max(n * self.params.report_psms, 1) * L.FEATURE_DTYPE.itemsize
            nb_c = max(n, 1) * 4
            pf, pc = C.c_void_p(), C.c_void_p()
            L.check(lib.sage_hip_host_alloc(nb_f, C.byref(pf)))
            L.check(lib.sage_hip_host_alloc(nb_c, C.byref(pc)))
            feats = np.frombuffer((C.c_char * nb_f).from_address(pf.value), dtype=L.FEATURE_DTYPE,
                                  count=n * self.params.report_psms)
            counts = np.frombuffer((C.c_char * nb_c).from_address(pc.value), dtype=np.uint32, count=n)
            self._pinned = (key, pf, pc, feats, counts)
        _, _, _, feats, counts = self._pinned
        return feats, counts

    def _free_pinned(self):
        cached = getattr(self, "_pinned", None)
        if cached is not None:
            lib = L.load()
            lib.sage_hip_host_free(cached[1])
            lib.sage_hip_host_free(cached[2])
            self._pinned = None

    def score_resident(self, dbatch: DeviceBatch):
        lib = L.load()
        feats, counts = self._alloc_out(dbatch.n)
        L.check(lib.sage_hip_score_resident(self._h, dbatch._h, feats.ctypes.data_as(C.c_void_p),
                                            L.as_ptr(counts, C.c_uint32)))
        return feats.reshape(dbatch.n, self.params.report_psms), counts

    def score(self, batch: SpectrumBatch, pinned_out: bool = True):
        """Vec<Feature> per spectrum: returns (features[n, report_psms], counts[n]).  Host arrays in, host arrays out, through
        the upload / score / download pipeline of sage_hip_score_batch.  pinned_out=False: plain (pageable) result arrays."""
        lib = L.load()
        if pinned_out:
            feats, counts = self._alloc_out(batch.n)
        else:
            feats = np.zeros(batch.n * self.params.report_psms, dtype=L.FEATURE_DTYPE)
            counts = np.zeros(batch.n, dtype=np.uint32)
        cb = batch.to_c()
        L.check(lib.sage_hip_score_batch(self._h, C.byref(cb), feats.ctypes.data_as(C.c_void_p),
                                         L.as_ptr(counts, C.c_uint32)))
        return feats.reshape(batch.n, self.params.report_psms), counts

    def annotate(self, dbatch: DeviceBatch, feats: np.ndarray, counts: np.ndarray):
        """Fragments (scoring.rs:152-161) of the PSMs that score_resident returned for this batch (annotate_matches).
        Returns (psm_off[n * report_psms + 1], dict of flat arrays); PSM r of spectrum i is slot i * report_psms + r."""
        lib = L.load()
        rp = self.params.report_psms
        feats = np.ascontiguousarray(feats).reshape(dbatch.n * rp)
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        valid = (np.arange(rp)[None, :] < counts[:, None]).reshape(-1)
        total = int(feats["matched_peaks"][valid].sum())
        off = np.zeros(dbatch.n * rp + 1, dtype=np.uint64)
        arr = dict(kinds=np.zeros(total, np.uint8), charges=np.zeros(total, np.int32), fragment_ordinals=np.zeros(total, np.int32),
                   intensities=np.zeros(total, np.float32), mz_calculated=np.zeros(total, np.float32),
                   mz_experimental=np.zeros(total, np.float32))
        fr = L.SageFragments(total, L.as_ptr(off, C.c_uint64), L.as_ptr(arr["kinds"], C.c_uint8),
                             arr["charges"].ctypes.data_as(C.POINTER(C.c_int32)),
                             arr["fragment_ordinals"].ctypes.data_as(C.POINTER(C.c_int32)),
                             L.as_ptr(arr["intensities"], C.c_float), L.as_ptr(arr["mz_calculated"], C.c_float),
                             L.as_ptr(arr["mz_experimental"], C.c_float))
        L.check(lib.sage_hip_annotate_resident(self._h, dbatch._h, feats.ctypes.data_as(C.c_void_p),
                                               L.as_ptr(counts, C.c_uint32), C.byref(fr)))
        return off, arr

    def quick_score(self, dbatch: DeviceBatch, prefilter_low_memory: bool, keep: Optional[np.ndarray] = None) -> np.ndarray:
        """Scorer::quick_score (scoring.rs:255-298) over the batch; `keep` ([n_peptides] u8) is OR-updated and returned."""
        lib = L.load()
        if keep is None:
            keep = np.zeros(self.db.host.n_peptides, dtype=np.uint8)
        assert keep.dtype == np.uint8 and len(keep) == self.db.host.n_peptides
        L.check(lib.sage_hip_quick_score_resident(self._h, dbatch._h, int(prefilter_low_memory), L.as_ptr(keep, C.c_uint8)))
        return keep

    def initial_hits(self, dbatch: DeviceBatch):
        lib = L.load()
        cap = max(50, 2 * self.params.report_psms)
        packed = np.zeros((dbatch.n, cap), dtype=np.uint64)
        ln = np.zeros(dbatch.n, dtype=np.uint32)
        mp = np.zeros(dbatch.n, dtype=np.uint64)
        sc = np.zeros(dbatch.n, dtype=np.uint64)
        L.check(lib.sage_hip_initial_hits(self._h, dbatch._h, L.as_ptr(packed, C.c_uint64), cap,
                                          L.as_ptr(ln, C.c_uint32), L.as_ptr(mp, C.c_uint64),
                                          L.as_ptr(sc, C.c_uint64)))
        return packed, ln, mp, sc

    def set_timing_interval(self, every: int):
        """Per-kernel HIP events on every `every`-th score_resident call only (0: never; default 1): sage_hip.h."""
        L.check(L.load().sage_hip_scorer_set_timing_interval(self._h, int(every)))

    def last_timing(self) -> dict:
        t = L.SageTiming()
        L.check(L.load().sage_hip_last_timing(self._h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in L.SageTiming._fields_}

    def close(self):
        if self._h:
            self._free_pinned()
            L.load().sage_hip_scorer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class RescoreResult:
    """Outputs of sage_hip_rescore, input order (Feature fields of scoring.rs:124-136) + the reference's output order."""
    discriminant_score: np.ndarray
    posterior_error: np.ndarray
    spectrum_q: np.ndarray
    peptide_q: np.ndarray
    protein_q: np.ndarray
    order: np.ndarray
    passing_spectrum: int
    passing_peptide: int
    passing_protein: int
    lda_fitted: bool
    coef: np.ndarray
    device_ms: float


def rescore(features: np.ndarray, precursor_tol: Tolerance, peptide_key, n_peptide_keys: int, protein_key,
            n_protein_keys: int, aligned_rt=None, delta_rt_model=None, delta_ims_model=None, device: int = 0) -> RescoreResult:
    """spectrum_fdr + picked_peptide + picked_protein (sage-cli runner.rs:536-541) over ALL Features of a run, on the
    device (rescore.hip).  `features`: 1-D array of FEATURE_DTYPE."""
    f = np.ascontiguousarray(features, dtype=L.FEATURE_DTYPE).reshape(-1)
    n = len(f)
    pk = np.ascontiguousarray(peptide_key, dtype=np.uint32)
    prk = np.ascontiguousarray(protein_key, dtype=np.uint32)
    assert len(pk) == n and len(prk) == n
    opt = [None if a is None else np.ascontiguousarray(a, dtype=np.float32) for a in (aligned_rt, delta_rt_model, delta_ims_model)]
    outs = [np.empty(n, np.float32) for _ in range(5)]
    order = np.empty(n, np.uint32)
    cin = L.SageRescoreInput(n, f.ctypes.data, *[None if a is None else L.as_ptr(a, C.c_float) for a in opt],
                             precursor_tol.to_c(), L.as_ptr(pk, C.c_uint32), n_peptide_keys, L.as_ptr(prk, C.c_uint32),
                             n_protein_keys)
    cout = L.SageRescoreOutput(*[L.as_ptr(a, C.c_float) for a in outs], L.as_ptr(order, C.c_uint32))
    L.check(L.load().sage_hip_rescore(device, C.byref(cin), C.byref(cout)))
    return RescoreResult(*outs, order, int(cout.passing_spectrum), int(cout.passing_peptide), int(cout.passing_protein),
                         bool(cout.lda_fitted), np.array(cout.coef[:], dtype=np.float64), float(cout.device_ms))


@dataclass
class RtPrediction:
    """Outputs of sage_hip_predict_rt, input order (Feature fields aligned_rt, predicted_rt, delta_rt_model, predicted_ims,
    delta_ims_model; spectrum_q of the poisson-sorted pass) + the per-file alignments."""
    spectrum_q: np.ndarray
    aligned_rt: np.ndarray
    predicted_rt: np.ndarray
    delta_rt_model: np.ndarray
    predicted_ims: np.ndarray
    delta_ims_model: np.ndarray
    alignments: np.ndarray  # [n_files] of (file_id, max_rt, slope, intercept)
    rt_fitted: bool
    ims_fitted: bool
    rt_r2: float
    ims_r2: float
    device_ms: float


ALIGNMENT_DTYPE = np.dtype([("file_id", "<u4"), ("max_rt", "<f4"), ("slope", "<f4"), ("intercept", "<f4")])


def predict_rt(features: np.ndarray, n_files: int, seq_off, seq, monoisotopic, device: int = 0) -> RtPrediction:
    """The predict_rt block of sage-cli (runner.rs:513-530) on the device: poisson-sorted q-values, global retention-time
    alignment, retention-time and ion-mobility linear models."""
    f = np.ascontiguousarray(features, dtype=L.FEATURE_DTYPE).reshape(-1)
    n = len(f)
    off = np.ascontiguousarray(seq_off, dtype=np.uint64)
    sq = np.ascontiguousarray(seq, dtype=np.uint8)
    mono = np.ascontiguousarray(monoisotopic, dtype=np.float32)
    assert len(off) == n + 1 and len(mono) == n
    outs = [np.empty(n, np.float32) for _ in range(6)]
    al = np.zeros(n_files, dtype=ALIGNMENT_DTYPE)
    cin = L.SageRtInput(n, f.ctypes.data, n_files, L.as_ptr(off, C.c_uint64), L.as_ptr(sq, C.c_uint8), L.as_ptr(mono, C.c_float))
    cout = L.SageRtOutput(*[L.as_ptr(a, C.c_float) for a in outs], C.cast(al.ctypes.data, C.POINTER(L.SageAlignment)))
    L.check(L.load().sage_hip_predict_rt(device, C.byref(cin), C.byref(cout)))
    return RtPrediction(*outs, al, bool(cout.rt_fitted), bool(cout.ims_fitted), float(cout.rt_r2), float(cout.ims_r2),
                        float(cout.device_ms))


def device_count() -> int:
    return int(L.load().sage_hip_device_count())
