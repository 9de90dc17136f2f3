"""Synthetic FASTA / MS2 generators for the BASELINE.json configs (recipes: SURVEY.md §8d).

The reference ships no generator; these are ours.  All randomness comes from
numpy.random.default_rng(seed) (PCG64), so workloads are reproducible across machines.
"""
import numpy as np

from . import _lib as _L
from .api import IndexedDatabase, RawSpectrum

# UniProtKB/Swiss-Prot background amino-acid frequencies (percent), 20 standard residues
_AA = "ACDEFGHIKLMNPQRSTVWY"
_FREQ = np.array([8.25, 1.38, 5.46, 6.72, 3.86, 7.07, 2.27, 5.91, 5.80, 9.65, 2.41, 4.06, 4.74, 3.93, 5.53, 6.65,
                  5.36, 6.86, 1.10, 2.92])
_FREQ = _FREQ / _FREQ.sum()

# f64 residue masses for the generator only (the engine's own f32 table is in csrc/host_db.cpp)
_MASS = {"A": 71.03711, "C": 103.00919, "D": 115.02694, "E": 129.04259, "F": 147.0684, "G": 57.02146,
         "H": 137.05891, "I": 113.08406, "K": 128.09496, "L": 113.08406, "M": 131.0405, "N": 114.04293,
         "P": 97.05276, "Q": 128.05858, "R": 156.1011, "S": 87.03203, "T": 101.04768, "V": 99.06841,
         "W": 186.07932, "Y": 163.06332, "U": 150.95363, "O": 237.14774}
_MASS_LUT = np.zeros(256)
for _k, _v in _MASS.items():
    _MASS_LUT[ord(_k)] = _v
PROTON = 1.0072764
H2O = 18.010565


def synthetic_fasta(n_proteins: int, seed: int, mu: float = 6.0, sigma: float = 0.6, lo: int = 50, hi: int = 3000) -> str:
    rng = np.random.default_rng(seed)
    lens = np.clip(rng.lognormal(mu, sigma, n_proteins).astype(np.int64), lo, hi)
    letters = np.frombuffer(_AA.encode(), dtype=np.uint8)
    out = []
    for i, n in enumerate(lens):
        seq = letters[rng.choice(len(_AA), size=int(n), p=_FREQ)].tobytes().decode()
        out.append(f">sp|SYN{i:06d}|SYN{i:06d}_SYNTH synthetic protein {i}\n")
        for j in range(0, len(seq), 60):
            out.append(seq[j:j + 60] + "\n")
    return "".join(out)


def paralog_fasta(n_families: int, copies: int, seed: int, mutation_rate: float = 0.01, il_swap_rate: float = 0.5,
                  repeat_frac: float = 0.15) -> str:
    """A proteome shaped like a real one where it hurts a k-select: families of `copies` paralogs — the family's founder with
    point mutations at `mutation_rate` of the residues, isoleucine / leucine swapped at `il_swap_rate` of their positions
    (identical masses: such peptides tie in every score) — and a `repeat_frac` of the families carrying a domain repeated in
    tandem.  Tryptic peptides are shared between paralogs, differ by one residue, or differ by nothing a mass spectrometer sees;
    the i.i.d. proteomes of synthetic_fasta have none of that."""
    rng = np.random.default_rng(seed)
    letters = np.frombuffer(_AA.encode(), dtype=np.uint8)
    base = synthetic_fasta(n_families, seed)
    founders = ["".join(b.split("\n")[1:]) for b in base.split(">")[1:]]
    out = []
    I, Lc = ord("I"), ord("L")
    for f, seq in enumerate(founders):
        a = np.frombuffer(seq.encode(), dtype=np.uint8).copy()
        if rng.random() < repeat_frac and len(a) > 120:  # a domain of 40-80 residues, two or three times in a row
            s0 = int(rng.integers(0, len(a) - 80))
            dom = a[s0:s0 + int(rng.integers(40, 80))]
            a = np.concatenate([a[:s0], np.tile(dom, int(rng.integers(2, 4))), a[s0:]])
        for c in range(copies):
            v = a.copy()
            if c:
                hit = rng.random(len(v)) < mutation_rate
                v[hit] = letters[rng.choice(len(_AA), size=int(hit.sum()), p=_FREQ)]
                il = ((v == I) | (v == Lc)) & (rng.random(len(v)) < il_swap_rate)
                v[il] = np.where(v[il] == I, Lc, I)
            sq = v.tobytes().decode()
            out.append(f">sp|FAM{f:05d}P{c}|FAM{f:05d}P{c}_SYNTH family {f} paralog {c}\n")
            for j in range(0, len(sq), 60):
                out.append(sq[j:j + 60] + "\n")
    return "".join(out)


def synthetic_spectra(db: IndexedDatabase, n_spectra: int, seed: int, noise_peaks: int = 80, pure_noise_frac: float = 0.10,
                      keep_prob: float = 0.5, ppm_sigma: float = 3.0, charges=((2, 0.6), (3, 0.3), (4, 0.1)),
                      annotate_charge: bool = True, mass_shift_frac: float = 0.0, chimeric: int = 1,
                      isolation_half_width: float = None, varmod_frac: float = None, varmod_residues: str = "M"):
    """Returns a list of RawSpectrum (centroided MS2).  Each spectrum is built from `chimeric` source
    peptides drawn uniformly from the target peptides of `db` (b/y ions, z=1 plus z=2 copies when the
    precursor charge is >= 3), fragment m/z error N(0, ppm_sigma), LogNormal intensities and uniform noise."""
    rng = np.random.default_rng(seed)
    targets = np.flatnonzero(db.decoy == 0)
    plain = modded = None
    if varmod_frac is not None:
        # a source peptide "carries a variable mod" when it has a terminal mod or a modified `varmod_residues` residue
        # (computed once per database: it walks every residue)
        cache = db.__dict__.setdefault("_synthetic_pools", {})
        if varmod_residues not in cache:
            flag = np.zeros(len(db.seq) + 1, dtype=np.int64)
            flag[:-1] = np.isin(db.seq, np.frombuffer(varmod_residues.encode(), np.uint8)) & (db.mods != 0)
            per_pep = np.add.reduceat(flag, db.seq_off[:-1].astype(np.int64)) if db.n_peptides else np.zeros(0, np.int64)
            is_mod = (per_pep > 0) | (np.nan_to_num(db.nterm) != 0) | (np.nan_to_num(db.cterm) != 0)
            cache[varmod_residues] = (targets[~is_mod[targets]], targets[is_mod[targets]])
        plain, modded = cache[varmod_residues]
        if len(plain) == 0 or len(modded) == 0:
            plain = modded = None
    zs = np.array([c for c, _ in charges])
    zp = np.array([p for _, p in charges])
    zp = zp / zp.sum()
    spectra = []
    seq_off = db.seq_off.astype(np.int64)
    for i in range(n_spectra):
        mzs, ints = [], []
        pure_noise = rng.random() < pure_noise_frac
        z = int(rng.choice(zs, p=zp))
        first_mz = None
        n_src = 1 if chimeric <= 1 else int(rng.integers(2, chimeric + 1))
        for s in range(n_src):
            if plain is not None:
                pool = modded if rng.random() < varmod_frac else plain
                pep = int(pool[rng.integers(len(pool))])
            else:
                pep = int(targets[rng.integers(len(targets))])
            a, b = seq_off[pep], seq_off[pep + 1]
            res = _MASS_LUT[db.seq[a:b]] + db.mods[a:b].astype(np.float64)
            nterm = float(db.nterm[pep]) if not np.isnan(db.nterm[pep]) else 0.0
            mono = float(db.pep_mono[pep])
            prec_mz = (mono + z * PROTON) / z * (1.0 + rng.normal(0.0, ppm_sigma) * 1e-6)
            if first_mz is None:
                first_mz = prec_mz
            elif isolation_half_width:
                # co-isolated peptide: keep only those that fall inside the isolation window
                if abs(prec_mz - first_mz) > isolation_half_width:
                    pass  # still add its fragments: chimeric spectra are messy by construction
            if pure_noise:
                continue
            bs = nterm + np.cumsum(res)[:-1]
            ys = mono - bs
            shift_site = None
            if mass_shift_frac and rng.random() < mass_shift_frac:
                delta = rng.uniform(-100.0, 400.0)
                site = int(rng.integers(0, len(res)))
                # an unknown modification of mass delta on residue `site`
                bs = bs + (np.arange(len(bs)) >= site) * delta
                ys = ys + (np.arange(len(ys)) < site) * delta
                if s == 0:
                    first_mz = (mono + delta + z * PROTON) / z * (1.0 + rng.normal(0.0, ppm_sigma) * 1e-6)
            frag = np.concatenate([bs, ys])
            cand = [frag + PROTON]
            if z >= 3:
                cand.append((frag + 2 * PROTON) / 2.0)
            cand = np.concatenate(cand)
            keep = rng.random(len(cand)) < keep_prob
            cand = cand[keep] * (1.0 + rng.normal(0.0, ppm_sigma, keep.sum()) * 1e-6)
            mzs.append(cand)
            ints.append(rng.lognormal(8.0, 1.2, len(cand)))
        mzs.append(rng.uniform(150.0, 1800.0, noise_peaks))
        ints.append(rng.lognormal(6.5, 1.0, noise_peaks))
        mz = np.concatenate(mzs)
        it = np.concatenate(ints)
        ok = (mz > 100.0) & (mz < 2500.0)
        mz, it = mz[ok], it[ok]
        order = np.argsort(mz, kind="stable")
        iso = (-isolation_half_width, isolation_half_width) if isolation_half_width else None
        spectra.append(RawSpectrum(mz[order].astype(np.float32), it[order].astype(np.float32), float(np.float32(first_mz)),
                                   z if annotate_charge else None, iso, scan_start_time=float(np.float32(i * 0.01)),
                                   file_id=0, id=f"scan={i + 1}"))
    return spectra


def synthetic_features(n: int, seed: int = 7, decoy_frac: float = 0.35, true_frac: float = 0.55, ppm: bool = True,
                       zero_ims: bool = False):
    """(For the post-search rescoring: sage_hip_rescore.)  n PSM records shaped like search output: decoys and false targets share one score distribution, true targets
    another.  Returns (features[FEATURE_DTYPE], peptide_key, n_peptide_keys, protein_key, n_protein_keys)."""
    rng = np.random.default_rng(seed)
    f = np.zeros(n, dtype=_L.FEATURE_DTYPE)
    decoy = rng.random(n) < decoy_frac
    true = ~decoy & (rng.random(n) < true_frac)
    f["spec_index"] = np.arange(n)
    f["label"] = np.where(decoy, -1, 1)
    f["rank"] = 1 + (rng.random(n) < 0.1)
    f["charge"] = rng.choice([2, 3, 4], size=n, p=[0.6, 0.3, 0.1])
    plen = rng.integers(7, 31, n)
    f["peptide_len"] = plen
    f["calcmass"] = (plen * 111.0 + rng.normal(0, 40, n)).astype(np.float32)
    err_ppm = np.where(true, rng.normal(0.5, 2.0, n), rng.uniform(-10, 10, n))
    f["expmass"] = (f["calcmass"].astype(np.float64) * (1 + err_ppm * 1e-6)).astype(np.float32)
    f["delta_mass"] = err_ppm.astype(np.float32) if ppm else (f["expmass"] - f["calcmass"])
    f["isotope_error"] = rng.choice([0.0, 1.0], size=n, p=[0.9, 0.1])
    f["average_ppm"] = np.where(true, rng.normal(0, 2, n), rng.normal(0, 5, n)).astype(np.float32)
    hs = np.where(true, rng.normal(32, 6, n), rng.normal(17, 3, n)).clip(2, None)
    f["hyperscore"] = hs
    f["delta_next"] = np.where(true, rng.gamma(4, 3, n), rng.gamma(1.2, 0.8, n))
    f["delta_best"] = np.where(f["rank"] == 1, 0.0, rng.gamma(1.5, 1.0, n))
    mp = np.where(true, rng.integers(8, 26, n), rng.integers(4, 9, n))
    f["matched_peaks"] = mp
    f["longest_b"] = np.minimum(mp // 3, plen - 1)
    f["longest_y"] = np.minimum(np.where(true, mp // 2, mp // 4), plen - 1)
    f["longest_y_pct"] = (f["longest_y"] / plen).astype(np.float32)
    f["matched_intensity_pct"] = np.where(true, rng.uniform(20, 70, n), rng.uniform(2, 25, n)).astype(np.float32)
    f["scored_candidates"] = rng.integers(10, 400, n)
    f["poisson"] = -np.where(true, rng.gamma(6, 1.5, n), rng.gamma(2, 1.0, n))
    f["missed_cleavages"] = rng.choice([0, 1, 2], size=n, p=[0.7, 0.25, 0.05])
    f["rt"] = rng.uniform(0, 1, n).astype(np.float32)
    # A constant column (ims == 0 without ion mobility) makes an exactly-zero row of the scatter matrix; the reference's
    # signed-maximum pivot search (gauss.rs:97-108) then skips a column whenever the other candidates are negative and the
    # fit "fails" for every epsilon — data dependent.  Tests that want a fitted model use a varying column.
    f["ims"] = 0.0 if zero_ims else rng.uniform(0.6, 1.4, n).astype(np.float32)
    f["ms2_intensity"] = rng.lognormal(10, 1, n).astype(np.float32)
    f["file_id"] = 0
    # competitions: a target/decoy pair shares a peptide key; several PSMs per peptide; ~20 % shared peptides
    n_pk = max(2, n // 3)
    pk = rng.integers(0, n_pk, n).astype(np.uint32)
    pk = np.unique(pk, return_inverse=True)[1].astype(np.uint32)  # dense
    n_pk = int(pk.max()) + 1
    f["peptide_idx"] = pk * 2 + decoy  # a target and its decoy are different peptides
    n_pr = max(2, n // 20)
    prot_of_pep = rng.integers(0, n_pr, n_pk)
    shared = rng.random(n_pk) < 0.2
    prk = prot_of_pep[pk].astype(np.int64)
    prk[shared[pk]] = -1
    used, inv = np.unique(prk[prk >= 0], return_inverse=True)
    out = np.full(n, 0xFFFFFFFF, dtype=np.uint32)
    out[prk >= 0] = inv.astype(np.uint32)
    return f, pk, n_pk, out, len(used)


def synthetic_rt_world(n: int, n_files: int = 3, seed: int = 9, with_ims: bool = True):
    """(For sage_hip_predict_rt.)  synthetic_features plus what the retention-time / mobility models need: a peptide sequence
    per PSM (PSMs sharing a peptide key share it), retention times that follow a linear function of the amino-acid
    composition, distorted per file by a linear map (what global_alignment undoes), ion mobilities that follow composition and
    charge.  Returns (features, seq_off, seq, monoisotopic)."""
    f, pk, n_pk, _, _ = synthetic_features(n, seed=seed)
    rng = np.random.default_rng(seed + 1)
    letters = np.frombuffer(_AA.encode(), dtype=np.uint8)
    lens = rng.integers(7, 31, n_pk)
    off_p = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    res = letters[rng.choice(len(_AA), size=int(off_p[-1]), p=_FREQ)]
    res[(off_p[1:] - 1).astype(np.int64)] = np.where(rng.random(n_pk) < 0.5, ord("K"), ord("R"))  # tryptic C-termini
    hydro = rng.normal(0, 1, 256)
    bulk = rng.normal(0, 1, 256)
    comp_h = np.add.reduceat(hydro[res], off_p[:-1].astype(np.int64))
    comp_b = np.add.reduceat(bulk[res], off_p[:-1].astype(np.int64))
    mass_p = (np.add.reduceat(_MASS_LUT[res], off_p[:-1].astype(np.int64)) + H2O).astype(np.float32)
    rt_p = comp_h - comp_h.min()
    rt_p = rt_p / rt_p.max()  # the "global" retention of a peptide in [0, 1]
    true = (f["label"] == 1) & (f["hyperscore"] > 24)
    file_id = rng.integers(0, n_files, n).astype(np.uint32)
    slope = rng.uniform(0.8, 1.1, n_files)
    icpt = rng.uniform(0.0, 0.08, n_files)
    run_len = rng.uniform(55.0, 120.0, n_files)  # minutes
    rel = np.where(true, rt_p[pk] + rng.normal(0, 0.01, n), rng.uniform(0, 1, n)).clip(0, 1)
    f["file_id"] = file_id
    f["rt"] = (((rel - icpt[file_id]) / slope[file_id]).clip(0, 1) * run_len[file_id]).astype(np.float32)
    z = f["charge"].astype(np.float64)
    ims_p = 0.6 + 0.25 * (comp_b - comp_b.min()) / (comp_b.max() - comp_b.min())
    f["ims"] = (np.where(true, ims_p[pk] + 0.12 * (z - 2) + rng.normal(0, 0.01, n), rng.uniform(0.5, 1.4, n))
                if with_ims else np.zeros(n)).astype(np.float32)
    f["peptide_len"] = lens[pk]
    f["calcmass"] = mass_p[pk]
    # per-PSM peptide data
    seq_off = np.concatenate([[0], np.cumsum(lens[pk])]).astype(np.uint64)
    starts = off_p[:-1][pk].astype(np.int64)
    idx = np.repeat(starts - seq_off[:-1].astype(np.int64), lens[pk]) + np.arange(int(seq_off[-1]))
    return f, seq_off, res[idx].copy(), mass_p[pk].copy()
