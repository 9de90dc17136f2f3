"""The BASELINE.json configurations (recipes: SURVEY.md §8d) as data, shared by bench.py and the config-scale
parity tests (tests/test_gpu_config_scale.py), so that what is measured and what is checked is the same workload.

`spectra` is the size the configuration names; a rank of an N-GPU run scores its shard of it (bench.py).  bench.py checks ALL
of a run's spectra against the oracle; `cpu_sample` is the prefix its work counters (SURVEY 8d's algorithmic bytes) are taken on,
`cpu_time_sample` (default: the same) the ONE prefix the CPU baseline is timed on at every thread count.
"""
from typing import List, Optional

_ENZ1 = dict(missed_cleavages=1, min_len=5, max_len=50, cleave_at="KR", restrict="P")
_ENZ2 = dict(missed_cleavages=2, min_len=5, max_len=50, cleave_at="KR", restrict="P")
_DB = dict(bucket_size=8192, peptide_min_mass=500.0, peptide_max_mass=5000.0, static_mods={"C": 57.0215}, generate_decoys=True)

CONFIGS = {
    "C2": dict(name="C2: 50k synthetic MS2 x yeast-like tryptic digest, ±10 ppm narrow search", proteins=6000,
               fasta_seed=1001, spectra=50000, spectra_seed=2001, db=dict(_DB, enzyme=_ENZ1),
               scorer=dict(), spectra_kwargs=dict(), cpu_sample=50000, cpu_time_sample=32768,
               metric="spectra/sec (whole node), fragment-index search-and-score, narrow search"),
    "C3": dict(name="C3: 500k synthetic MS2 x human-like tryptic digest + 2 variable mods (M+15.9949, protein N-term "
                    "+42.0106), ±10 ppm narrow search", proteins=20400,
               fasta_seed=1002, spectra=500000, spectra_seed=2002,
               db=dict(_DB, enzyme=_ENZ1, variable_mods={"M": [15.9949], "[": [42.010565]}, max_variable_mods=2),
               scorer=dict(), spectra_kwargs=dict(varmod_frac=0.15), cpu_sample=65536, cpu_time_sample=32768,
               metric="spectra/sec (whole node), fragment-index search-and-score, human tryptic narrow search"),
    "C4": dict(name="C4: 100k synthetic MS2 x human-like tryptic digest (2 missed cleavages, variable M+15.9949), open "
                    "search da[-500,100]", proteins=20400,
               fasta_seed=1002, spectra=100000, spectra_seed=2004,
               db=dict(_DB, enzyme=_ENZ2, variable_mods={"M": [15.9949]}, max_variable_mods=2),
               scorer=dict(precursor_tol=("da", -500.0, 100.0)), spectra_kwargs=dict(mass_shift_frac=0.3),
               cpu_sample=2048,
               metric="spectra/sec (whole node), fragment-index search-and-score, open search"),
    "C5": dict(name="C5: 200k chimeric synthetic MS2 (2-3 peptides per 12 Th isolation window, no charge annotation) x "
                    "human-like tryptic digest + 2 variable mods, wide_window + chimera, report_psms 5",
               proteins=20400, fasta_seed=1002, spectra=200000, spectra_seed=2005,
               db=dict(_DB, enzyme=_ENZ1, variable_mods={"M": [15.9949], "[": [42.010565]}, max_variable_mods=2),
               scorer=dict(wide_window=True, chimera=True, report_psms=5, min_precursor_charge=2, max_precursor_charge=4),
               spectra_kwargs=dict(chimeric=3, isolation_half_width=6.0, annotate_charge=False), cpu_sample=4096,
               metric="spectra/sec (whole node), fragment-index search-and-score, chimeric wide-window search"),
    # not a BASELINE.json configuration: C3's digest and search over a proteome of paralog families (synthetic.paralog_fasta:
    # shared and near-identical peptides, I/L twins, tandem repeats), where equal hyperscores at the reported rank are common —
    # the parity suite's tie-rich case (tests/test_gpu_config_scale.py) and the retry-pass stress of scripts/ab_env.py
    "C3T": dict(name="C3T: C3's search over 5100 paralog families x 4 (shared / near-identical / I-L-twin peptides)", proteins=5100,
                fasta_seed=1012, spectra=500000, spectra_seed=2012, fasta="paralog", paralog_copies=4,
                db=dict(_DB, enzyme=_ENZ1, variable_mods={"M": [15.9949], "[": [42.010565]}, max_variable_mods=2),
                scorer=dict(), spectra_kwargs=dict(varmod_frac=0.15), cpu_sample=16384,
                metric="spectra/sec (whole node), fragment-index search-and-score, human tryptic narrow search, tie-rich proteome"),
}
DEFAULT_CONFIG = "C3"  # BASELINE.json's metric is quoted on the human tryptic narrow search; it fits one GPU

# synthetic spectra are generated in chunks of this many, chunk c from seed `spectra_seed * 1000 + c`: any rank can
# produce exactly its own contiguous shard of THE workload without generating the rest
SPECTRA_CHUNK = 4096


def scorer_params(cfg):
    from .api import ScorerParams, Tolerance
    kw = dict(cfg["scorer"])
    for k in ("precursor_tol", "fragment_tol"):
        if k in kw:
            kw[k] = Tolerance(*kw[k])
    return ScorerParams(**kw)


def build_host_db(cfg, proteins: Optional[int] = None, peptides_only: bool = False):
    from .api import DatabaseParameters
    from .synthetic import paralog_fasta, synthetic_fasta
    if cfg.get("fasta") == "paralog":
        fasta = paralog_fasta(proteins or cfg["proteins"], cfg["paralog_copies"], cfg["fasta_seed"])
    else:
        fasta = synthetic_fasta(proteins or cfg["proteins"], cfg["fasta_seed"])
    return DatabaseParameters(**cfg["db"]).build(fasta, peptides_only=peptides_only)


def processed_spectra(cfg, host, begin: int, end: int, total: Optional[int] = None):
    """Spectra [begin, end) of the configuration's synthetic run, preprocessed as the CLI defaults do (max_peaks 150,
    deisotope, input.rs:366,371).  Returns a list of (global index, ProcessedSpectrum); spectra with fewer than
    min_peaks = 15 peaks (runner.rs:313) are dropped, exactly as the reference drops them before scoring."""
    from .api import SpectrumProcessor
    from .synthetic import synthetic_spectra
    total = total if total is not None else cfg["spectra"]
    end = min(end, total)
    sp = SpectrumProcessor(150, True, 0.0)
    out: List = []
    c0, c1 = begin // SPECTRA_CHUNK, (max(end, begin + 1) - 1) // SPECTRA_CHUNK
    for c in range(c0, c1 + 1):
        lo = c * SPECTRA_CHUNK
        n = min(SPECTRA_CHUNK, total - lo)
        if n <= 0:
            break
        raws = synthetic_spectra(host, n, cfg["spectra_seed"] * 1000 + c, **cfg["spectra_kwargs"])
        for j, r in enumerate(raws):
            g = lo + j
            if g < begin or g >= end:
                continue
            p = sp.process(r)
            if len(p.masses) >= 15:
                out.append((g, p))
    return out


def workload_batch(cfg, host, begin: int, end: int, total: Optional[int] = None):
    """SpectrumBatch of spectra [begin, end) (after the min_peaks filter) + their global indices."""
    import numpy as np
    from .api import SpectrumBatch
    items = processed_spectra(cfg, host, begin, end, total)
    return SpectrumBatch.from_spectra([p for _, p in items]), np.array([g for g, _ in items], dtype=np.int64)
