"""Randomised scorer configurations: the HIP path (through the C ABI) against the oracle, field for field.

The hand-written parity cases (test_gpu_parity.py) each aim at one code path.  Here the 15 `Scorer` fields (scoring.rs:210-232) and the
database's ion kinds / min_ion_index / enzyme / bucket size are drawn TOGETHER from a seeded generator, so that combinations nobody
thought of meet — an asymmetric Da fragment tolerance under a chimeric wide-window search with five isotope errors, six ion kinds with
fragment charge 4 behind an overridden precursor charge, a one-sided precursor window that selects nothing.  Every case goes
through World.check: the preliminary lists (heap layout), every Feature field, a second step on the same handle, and the
host-to-host entry point.  The seeds are fixed: the suite is deterministic, and a case that ever fails is reproduced by its id."""
import os

import numpy as np
import pytest

from sage_amd.api import DatabaseParameters, ScorerParams, SpectrumBatch, Tolerance
from sage_amd.synthetic import synthetic_fasta
from test_gpu_parity import World

pytestmark = pytest.mark.gpu

WORLDS = {
    # (database parameters, synthetic_spectra arguments, spectra, peaks kept per spectrum)
    "by": (dict(bucket_size=2048, enzyme=dict(missed_cleavages=1, cleave_at="KR", restrict="P"), static_mods={"C": 57.0215},
                variable_mods={"M": [15.9949]}), {}, 240, 150),
    "abcxyz": (dict(bucket_size=512, enzyme=dict(missed_cleavages=2, cleave_at="KR", restrict="P", min_len=6, max_len=40),
                    ion_kinds=["a", "b", "c", "x", "y", "z"], min_ion_index=1, static_mods={"C": 57.0215},
                    variable_mods={"M": [15.9949], "S": [79.9663]}, max_variable_mods=2),
               dict(chimeric=2, isolation_half_width=4.0), 160, 90),
    "y_only": (dict(bucket_size=8192, enzyme=dict(missed_cleavages=0, cleave_at="FWYL", restrict=None, c_terminal=False, min_len=5, max_len=30),
                    ion_kinds=["y"], min_ion_index=3, generate_decoys=False),
               dict(annotate_charge=False, noise_peaks=30, keep_prob=0.8), 160, 40),
}
# The suite's cases: 24 per world, salt 2026.  A campaign (scripts/gpu_fuzz_campaign.sh, filed as profiles/r06_fuzz_campaign.txt) runs
# the same test over other cases: SAGE_FUZZ_CASES per world, drawn under SAGE_FUZZ_SALT.
CASES_PER_WORLD = int(os.environ.get("SAGE_FUZZ_CASES", "24"))
SALT = int(os.environ.get("SAGE_FUZZ_SALT", "2026"))


def _tolerance(rng, fragment):
    kind = rng.choice(["ppm", "ppm", "da", "pct"] if fragment else ["ppm", "ppm", "da", "da", "pct"])
    shape = rng.choice(["sym", "sym", "asym", "one_sided"])
    if kind == "ppm":
        a, b = rng.uniform(3.0, 60.0, 2)
    elif kind == "da":
        a, b = rng.uniform(0.005, 0.4, 2) if fragment else rng.uniform(0.01, 6.0, 2)
    else:  # percent of the centre (mass.rs:24)
        a, b = rng.uniform(0.0002, 0.004, 2) if fragment else rng.uniform(0.0005, 0.05, 2)
    if shape == "sym":
        lo, hi = -a, a
    elif shape == "asym":
        lo, hi = -a, b
    else:  # the window lies beside the centre
        lo, hi = (min(a, b) * 0.2, max(a, b)) if rng.random() < 0.5 else (-max(a, b), -min(a, b) * 0.2)
    return Tolerance(str(kind), float(np.float32(lo)), float(np.float32(hi)))


def _params(rng):
    p = ScorerParams()
    p.fragment_tol = _tolerance(rng, True)
    open_search = rng.random() < 0.12
    if open_search:
        w = float(rng.uniform(40.0, 250.0))
        p.precursor_tol = Tolerance("da", -w, float(rng.uniform(0.3, 1.0)) * w)
    else:
        p.precursor_tol = _tolerance(rng, False)
    p.min_matched_peaks = int(rng.integers(0, 7))
    lo = int(rng.integers(-2, 2))
    p.min_isotope_err, p.max_isotope_err = lo, lo + int(rng.integers(0, 4))
    z0 = int(rng.integers(1, 4))
    p.min_precursor_charge, p.max_precursor_charge = z0, z0 + int(rng.integers(0, 4))
    p.override_precursor_charge = bool(rng.random() < 0.25)
    p.max_fragment_charge = [None, None, 1, 2, 3, 4][int(rng.integers(0, 6))]
    p.chimera = bool(rng.random() < 0.3)
    p.report_psms = int(rng.choice([1, 1, 1, 2, 3, 5, 8, 20, 40]))
    p.wide_window = bool(rng.random() < 0.2) and not open_search
    p.score_type = "OpenMSHyperScore" if rng.random() < 0.2 else "SageHyperScore"
    return p, open_search


def _coarse_peaks(batch, rng, zeros):
    """The same spectra as a coarse instrument would report them: intensities on a few levels (so that equally intense peaks meet
    inside a fragment window — spectrum.rs:147-157 keeps the LAST of them — and candidates tie in summed intensity), every ~12th
    peak's mass repeated by its neighbour (equal masses side by side: the order stays ascending), optionally some intensities at
    zero.  total_ion_current is re-summed in peak order, as SpectrumProcessor does."""
    it = batch.intensities.copy()
    top = float(it.max()) if len(it) else 1.0
    levels = int(rng.integers(3, 9))
    it = (np.ceil(it / np.float32(top) * levels) * np.float32(top / levels)).astype(np.float32)
    if zeros:
        it[rng.random(len(it)) < 0.1] = 0.0
    m = batch.masses.copy()
    off = batch.peak_off.astype(np.int64)
    dup = np.flatnonzero(rng.random(len(m)) < 0.08)
    last = np.zeros(len(m) + 1, bool)
    last[off[1:] - 1] = True  # (a spectrum's last peak has no right neighbour inside the spectrum)
    dup = dup[~last[dup]]
    m[dup + 1] = m[dup]
    # (in peak order, f32 — the running sum of spectrum.rs:396-400; a cumulative sum accumulates sequentially)
    tic = np.array([np.cumsum(it[off[i]:off[i + 1]], dtype=np.float32)[-1] if off[i + 1] > off[i] else 0.0 for i in range(batch.n)], np.float32)
    return SpectrumBatch(batch.peak_off, m, it, batch.precursor_mz, batch.precursor_charge, tic, batch.isolation_lo, batch.isolation_hi,
                         batch.scan_start_time, batch.inverse_ion_mobility, batch.file_id)


@pytest.fixture(scope="module", params=sorted(WORLDS))
def world(request, gpu_required):
    db, spectra, n, peaks = WORLDS[request.param]
    fasta = synthetic_fasta(180, seed=31 + sorted(WORLDS).index(request.param))
    w = World(fasta, DatabaseParameters(**db), spectra, n, seed=41 + sorted(WORLDS).index(request.param), max_peaks=peaks)
    w.name = request.param
    return w


@pytest.mark.parametrize("case", range(CASES_PER_WORLD))
def test_random_scorer_configuration(world, case):
    rng = np.random.default_rng([sorted(WORLDS).index(world.name), case, SALT])
    params, open_search = _params(rng)
    batch = world.batch
    if open_search or params.wide_window or params.report_psms >= 20:
        batch = batch.subset(np.arange(int(rng.integers(0, 4)), batch.n, 4))  # (the oracle scores every in-window peptide)
    if rng.random() < 0.3:  # unknown precursor charges: the scorer's charge range instead (scoring.rs:437-447)
        batch = SpectrumBatch(batch.peak_off, batch.masses, batch.intensities, batch.precursor_mz, np.zeros(batch.n, np.uint8),
                              batch.total_ion_current, batch.isolation_lo, batch.isolation_hi, batch.scan_start_time,
                              batch.inverse_ion_mobility, batch.file_id)
    how = rng.random()
    if how < 0.35:
        batch = _coarse_peaks(batch, rng, zeros=how < 0.1)
    world.check(params, f"{world.name}/{case}: {params}", batch=batch)
