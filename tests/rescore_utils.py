"""Synthetic Feature tables for the post-search rescoring tests (test infrastructure)."""
import numpy as np

from sage_amd import _lib as L


def synthetic_features(n: int, seed: int = 7, decoy_frac: float = 0.35, true_frac: float = 0.55, ppm: bool = True,
                       zero_ims: bool = False):
    """n PSM records shaped like search output: decoys and false targets share one score distribution, true targets
    another.  Returns (features[FEATURE_DTYPE], peptide_key, n_peptide_keys, protein_key, n_protein_keys)."""
    rng = np.random.default_rng(seed)
    f = np.zeros(n, dtype=L.FEATURE_DTYPE)
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
    used = np.unique(prk[prk >= 0])
    remap = {int(v): i for i, v in enumerate(used)}
    prk = np.array([remap[int(v)] if v >= 0 else 0xFFFFFFFF for v in prk], dtype=np.uint32)
    return f, pk, n_pk, prk, len(used)
