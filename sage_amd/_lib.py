"""ctypes binding of libsage_hip.so (include/sage_hip.h).  No fallback: if the library is missing it
is built; if it cannot be built or loaded, importing the scoring API raises."""
import ctypes as C
import os

import numpy as np

from . import build as _build

c_float_p = C.POINTER(C.c_float)
c_u8_p = C.POINTER(C.c_uint8)
c_u32_p = C.POINTER(C.c_uint32)
c_u64_p = C.POINTER(C.c_uint64)


class SageTolerance(C.Structure):
    _fields_ = [("kind", C.c_int32), ("lo", C.c_float), ("hi", C.c_float)]


class SageTheoretical(C.Structure):
    _fields_ = [("peptide_index", C.c_uint32), ("fragment_mz", C.c_float)]


class SageDbParams(C.Structure):
    _fields_ = [
        ("bucket_size", C.c_uint64),
        ("missed_cleavages", C.c_int32),
        ("min_len", C.c_int32),
        ("max_len", C.c_int32),
        ("cleave_at", C.c_char_p),
        ("restrict_", C.c_char_p),
        ("c_terminal", C.c_int32),
        ("semi_enzymatic", C.c_int32),
        ("enzyme_present", C.c_int32),
        ("peptide_min_mass", C.c_float),
        ("peptide_max_mass", C.c_float),
        ("ion_kinds", c_u8_p),
        ("n_ion_kinds", C.c_uint32),
        ("min_ion_index", C.c_uint64),
        ("static_mod_keys", C.POINTER(C.c_char_p)),
        ("static_mod_masses", c_float_p),
        ("n_static_mods", C.c_uint32),
        ("var_mod_keys", C.POINTER(C.c_char_p)),
        ("var_mod_masses", c_float_p),
        ("n_var_mods", C.c_uint32),
        ("max_variable_mods", C.c_uint64),
        ("decoy_tag", C.c_char_p),
        ("generate_decoys", C.c_int32),
        ("peptides_only", C.c_int32),
    ]


class SageDbView(C.Structure):
    _fields_ = [
        ("fragments", C.POINTER(SageTheoretical)),
        ("n_fragments", C.c_uint64),
        ("min_value", c_float_p),
        ("n_buckets", C.c_uint64),
        ("bucket_size", C.c_uint64),
        ("pep_mono", c_float_p),
        ("seq_off", c_u64_p),
        ("seq", c_u8_p),
        ("mods", c_float_p),
        ("nterm", c_float_p),
        ("cterm", c_float_p),
        ("decoy", c_u8_p),
        ("missed_cleavages", c_u8_p),
        ("n_peptides", C.c_uint64),
        ("ion_kinds", c_u8_p),
        ("n_ion_kinds", C.c_uint32),
        ("min_ion_index", C.c_uint64),
    ]


class SageScorerParams(C.Structure):
    _fields_ = [
        ("precursor_tol", SageTolerance),
        ("fragment_tol", SageTolerance),
        ("min_matched_peaks", C.c_uint16),
        ("min_isotope_err", C.c_int8),
        ("max_isotope_err", C.c_int8),
        ("min_precursor_charge", C.c_uint8),
        ("max_precursor_charge", C.c_uint8),
        ("override_precursor_charge", C.c_uint8),
        ("chimera", C.c_uint8),
        ("max_fragment_charge", C.c_int16),
        ("wide_window", C.c_uint8),
        ("annotate_matches", C.c_uint8),
        ("report_psms", C.c_uint32),
        ("score_type", C.c_int32),
    ]


class SageSpectrumBatch(C.Structure):
    _fields_ = [
        ("n_spectra", C.c_uint32),
        ("peak_off", c_u64_p),
        ("masses", c_float_p),
        ("intensities", c_float_p),
        ("precursor_mz", c_float_p),
        ("precursor_charge", c_u8_p),
        ("isolation_lo", c_float_p),
        ("isolation_hi", c_float_p),
        ("total_ion_current", c_float_p),
        ("scan_start_time", c_float_p),
        ("inverse_ion_mobility", c_float_p),
        ("file_id", c_u32_p),
    ]


class SageFragments(C.Structure):
    _fields_ = [
        ("capacity", C.c_uint64),
        ("psm_off", c_u64_p),
        ("kinds", c_u8_p),
        ("charges", C.POINTER(C.c_int32)),
        ("fragment_ordinals", C.POINTER(C.c_int32)),
        ("intensities", c_float_p),
        ("mz_calculated", c_float_p),
        ("mz_experimental", c_float_p),
    ]


class SageRawBatch(C.Structure):
    _fields_ = [
        ("n_spectra", C.c_uint32),
        ("peak_off", c_u64_p),
        ("mz", c_float_p),
        ("intensities", c_float_p),
        ("precursor_mz", c_float_p),
        ("precursor_charge", c_u8_p),
        ("isolation_lo", c_float_p),
        ("isolation_hi", c_float_p),
        ("scan_start_time", c_float_p),
        ("inverse_ion_mobility", c_float_p),
        ("file_id", c_u32_p),
    ]


class SageTiming(C.Structure):
    _fields_ = [
        ("prelim_ms", C.c_float),
        ("rescore_ms", C.c_float),
        ("total_ms", C.c_float),
        ("n_launches", C.c_uint32),
        ("n_wide", C.c_uint32),
        ("arena_entries", C.c_uint32),
        ("n_retry", C.c_uint32),
        ("retry_ms", C.c_float),
        ("n_tied", C.c_uint32),
        ("n_ways", C.c_uint32),
    ]


# numpy mirror of SageFeature (120 bytes)
FEATURE_DTYPE = np.dtype(
    [
        ("spec_index", "<u4"), ("peptide_idx", "<u4"), ("rank", "<u4"), ("label", "<i4"),
        ("expmass", "<f4"), ("calcmass", "<f4"), ("rt", "<f4"), ("ims", "<f4"), ("delta_mass", "<f4"),
        ("isotope_error", "<f4"), ("average_ppm", "<f4"), ("longest_y_pct", "<f4"),
        ("matched_intensity_pct", "<f4"), ("ms2_intensity", "<f4"),
        ("hyperscore", "<f8"), ("delta_next", "<f8"), ("delta_best", "<f8"), ("poisson", "<f8"),
        ("matched_peaks", "<u4"), ("longest_b", "<u4"), ("longest_y", "<u4"), ("scored_candidates", "<u4"),
        ("peptide_len", "<u4"), ("file_id", "<u4"), ("charge", "u1"), ("missed_cleavages", "u1"),
        ("pad", "u1", (6,)),
    ],
    align=False,
)
assert FEATURE_DTYPE.itemsize == 120

THEORETICAL_DTYPE = np.dtype([("peptide_index", "<u4"), ("fragment_mz", "<f4")])

ION_KINDS = {"a": 0, "b": 1, "c": 2, "x": 3, "y": 4, "z": 5}
TOL_KINDS = {"ppm": 0, "pct": 1, "da": 2}
SCORE_TYPES = {"SageHyperScore": 0, "OpenMSHyperScore": 1}


class SageRescoreInput(C.Structure):
    _fields_ = [("n", C.c_uint64), ("features", C.c_void_p), ("aligned_rt", c_float_p), ("delta_rt_model", c_float_p),
                ("delta_ims_model", c_float_p), ("precursor_tol", SageTolerance), ("peptide_key", c_u32_p),
                ("n_peptide_keys", C.c_uint32), ("protein_key", c_u32_p), ("n_protein_keys", C.c_uint32)]


class SageRescoreOutput(C.Structure):
    _fields_ = [("discriminant_score", c_float_p), ("posterior_error", c_float_p), ("spectrum_q", c_float_p),
                ("peptide_q", c_float_p), ("protein_q", c_float_p), ("order", c_u32_p), ("passing_spectrum", C.c_uint64),
                ("passing_peptide", C.c_uint64), ("passing_protein", C.c_uint64), ("lda_fitted", C.c_int32),
                ("coef", C.c_double * 20), ("device_ms", C.c_float)]


class SageAlignment(C.Structure):
    _fields_ = [("file_id", C.c_uint32), ("max_rt", C.c_float), ("slope", C.c_float), ("intercept", C.c_float)]


class SageRtInput(C.Structure):
    _fields_ = [("n", C.c_uint64), ("features", C.c_void_p), ("n_files", C.c_uint32), ("seq_off", c_u64_p), ("seq", c_u8_p),
                ("monoisotopic", c_float_p)]


class SageRtOutput(C.Structure):
    _fields_ = [("spectrum_q", c_float_p), ("aligned_rt", c_float_p), ("predicted_rt", c_float_p), ("delta_rt_model", c_float_p),
                ("predicted_ims", c_float_p), ("delta_ims_model", c_float_p), ("alignments", C.POINTER(SageAlignment)),
                ("rt_fitted", C.c_int32), ("ims_fitted", C.c_int32), ("rt_r2", C.c_double), ("ims_r2", C.c_double),
                ("device_ms", C.c_float)]


class SagePostColumns(C.Structure):
    _fields_ = [(k, c_float_p) for k in ("discriminant_score", "posterior_error", "spectrum_q", "peptide_q", "protein_q",
                                         "aligned_rt", "predicted_rt", "delta_rt_model", "predicted_ims", "delta_ims_model")]


class SageHipError(RuntimeError):
    pass


_lib = None


def lib_path():
    return _build.LIB


def load():
    """Load (building first if needed) libsage_hip.so.  Raises if that is impossible."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_build.LIB) or (os.environ.get("SAGE_HIP_REBUILD") == "1"):
        _build.build(verbose=False)
    lib = C.CDLL(_build.LIB)
    vp = C.c_void_p
    sig = {
        "sage_hip_last_error": (C.c_char_p, []),
        "sage_hip_abi_version": (C.c_int, []),
        "sage_hip_hostdb_build": (C.c_int, [C.c_char_p, C.POINTER(SageDbParams), C.POINTER(vp)]),
        "sage_hip_hostdb_free": (None, [vp]),
        "sage_hip_hostdb_view": (C.c_int, [vp, C.POINTER(SageDbView)]),
        "sage_hip_hostdb_peptide_string": (C.c_uint64, [vp, C.c_uint64, C.c_char_p, C.c_uint64]),
        "sage_hip_hostdb_peptide_proteins": (C.c_uint64, [vp, C.c_uint64, C.c_char_p, C.c_uint64]),
        "sage_hip_hostdb_peptide_info": (C.c_int, [vp, C.c_uint64, c_u32_p, c_u8_p]),
        "sage_hip_process_ms2": (C.c_uint64, [C.c_uint64, C.c_int, C.c_float, c_float_p, c_float_p, C.c_uint64,
                                              C.c_uint8, c_float_p, c_float_p, c_float_p]),
        "sage_hip_device_count": (C.c_int, []),
        "sage_hip_db_create": (C.c_int, [C.POINTER(SageDbView), C.c_int, C.POINTER(vp)]),
        "sage_hip_db_destroy": (None, [vp]),
        "sage_hip_db_device_bytes": (C.c_uint64, [vp]),
        "sage_hip_scorer_create": (C.c_int, [vp, C.POINTER(SageScorerParams), C.POINTER(vp)]),
        "sage_hip_scorer_destroy": (None, [vp]),
        "sage_hip_scorer_clone": (C.c_int, [vp, C.POINTER(vp)]),
        "sage_hip_score_batch": (C.c_int, [vp, C.POINTER(SageSpectrumBatch), vp, c_u32_p]),
        "sage_hip_batch_upload": (C.c_int, [vp, C.POINTER(SageSpectrumBatch), C.POINTER(vp)]),
        "sage_hip_batch_free": (None, [vp]),
        "sage_hip_batch_process_upload": (C.c_int, [vp, C.POINTER(SageRawBatch), C.c_uint64, C.c_int, C.c_float, C.c_uint32,
                                                    C.POINTER(vp), c_u32_p]),
        "sage_hip_batch_download": (C.c_int, [vp, c_u64_p, c_float_p, c_float_p, c_float_p]),
        "sage_hip_score_resident": (C.c_int, [vp, vp, vp, c_u32_p]),
        "sage_hip_initial_hits": (C.c_int, [vp, vp, c_u64_p, C.c_uint32, c_u32_p, c_u64_p, c_u64_p]),
        "sage_hip_last_timing": (C.c_int, [vp, C.POINTER(SageTiming)]),
        "sage_hip_scorer_set_timing_interval": (C.c_int, [vp, C.c_uint32]),
        "sage_hip_annotate_resident": (C.c_int, [vp, vp, vp, c_u32_p, C.POINTER(SageFragments)]),
        "sage_hip_quick_score_resident": (C.c_int, [vp, vp, C.c_int, c_u8_p]),
        "sage_hip_debug_phase_cycles": (C.c_int, [vp, c_u64_p]),
        "sage_hip_host_alloc": (C.c_int, [C.c_uint64, C.POINTER(vp)]),
        "sage_hip_host_free": (None, [vp]),
        "sage_hip_rescore": (C.c_int, [C.c_int, C.POINTER(SageRescoreInput), C.POINTER(SageRescoreOutput)]),
        "sage_hip_hostdb_competition_keys": (C.c_int, [vp, c_u32_p, C.c_uint64, c_u32_p, c_u32_p, c_u32_p, c_u32_p]),
        "sage_hip_predict_rt": (C.c_int, [C.c_int, C.POINTER(SageRtInput), C.POINTER(SageRtOutput)]),
        "sage_hip_hostdb_feature_peptides": (C.c_int, [vp, c_u32_p, C.c_uint64, c_u64_p, c_u8_p, c_float_p]),
        "sage_hip_mzml_read": (C.c_int, [C.c_char_p, C.c_uint32, C.c_int, C.POINTER(vp)]),
        "sage_hip_mzml_view": (C.c_int, [vp, C.POINTER(SageRawBatch)]),
        "sage_hip_mzml_check_searchable": (C.c_int, [vp]),
        "sage_hip_mzml_spectrum_id": (C.c_char_p, [vp, C.c_uint64]),
        "sage_hip_mzml_free": (None, [vp]),
        "sage_hip_write_results": (C.c_int, [C.c_char_p, C.c_int, vp, vp, C.c_uint64, c_u64_p, c_u64_p, C.POINTER(C.c_char_p),
                                             C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(SagePostColumns)]),
        "sage_hip_fasta_num_targets": (C.c_int, [C.c_char_p, C.POINTER(SageDbParams), c_u64_p]),
        "sage_hip_prefilter_chunk_size": (C.c_int, [C.c_char_p, C.POINTER(SageDbParams), C.c_uint64, c_u64_p]),
        "sage_hip_hostdb_build_chunk": (C.c_int, [C.c_char_p, C.POINTER(SageDbParams), C.c_uint64, C.c_uint64, C.POINTER(vp)]),
        "sage_hip_hostdb_merge_kept": (C.c_int, [C.POINTER(vp), C.POINTER(c_u8_p), C.c_uint32, C.POINTER(SageDbParams),
                                                 C.POINTER(vp)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


EXPORTED_SYMBOLS = [
    "sage_hip_last_error", "sage_hip_abi_version", "sage_hip_hostdb_build", "sage_hip_hostdb_free",
    "sage_hip_hostdb_view", "sage_hip_hostdb_peptide_string", "sage_hip_hostdb_peptide_proteins",
    "sage_hip_hostdb_peptide_info", "sage_hip_process_ms2", "sage_hip_device_count", "sage_hip_db_create", "sage_hip_db_destroy",
    "sage_hip_db_device_bytes", "sage_hip_scorer_create", "sage_hip_scorer_destroy", "sage_hip_scorer_clone", "sage_hip_score_batch",
    "sage_hip_batch_upload", "sage_hip_batch_free", "sage_hip_batch_process_upload", "sage_hip_batch_download", "sage_hip_score_resident", "sage_hip_initial_hits",
    "sage_hip_last_timing", "sage_hip_scorer_set_timing_interval", "sage_hip_annotate_resident", "sage_hip_quick_score_resident", "sage_hip_debug_phase_cycles", "sage_hip_host_alloc", "sage_hip_host_free",
    "sage_hip_rescore", "sage_hip_hostdb_competition_keys", "sage_hip_fasta_num_targets", "sage_hip_prefilter_chunk_size",
    "sage_hip_hostdb_build_chunk", "sage_hip_hostdb_merge_kept", "sage_hip_predict_rt", "sage_hip_hostdb_feature_peptides",
    "sage_hip_write_results", "sage_hip_mzml_read", "sage_hip_mzml_view", "sage_hip_mzml_check_searchable", "sage_hip_mzml_spectrum_id", "sage_hip_mzml_free",
]


def check(rc):
    if rc != 0:
        msg = load().sage_hip_last_error()
        raise SageHipError(f"libsage_hip status {rc}: {msg.decode() if msg else ''}")


def as_ptr(arr, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype))
