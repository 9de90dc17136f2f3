"""A SECOND, independent reading of the scoring half of the hot path — TEST INFRASTRUCTURE.

Written from the reference's Rust alone (`/root/reference/crates/sage/src/scoring.rs:43-67, 170-201, 239-247, 322-462, 478-672,
675-793`, `spectrum.rs:134-159`, `database.rs:281-292, 402-425, 526-561`, `ion_series.rs:36-85`, `heap.rs:7-60`,
`mass.rs:5-7, 21-35, 47-57, 64-76`), without consulting `oracle/sage_oracle.cpp`: straight loops over np.float32 scalars, no
fragment index (every peptide of the precursor window is matched by brute force against its own regenerated ion series), Python
tuples for `PreScore`'s derived `Ord`.  `tests/test_scoring_second_reading.py` holds the C++ oracle to it — VERDICT r05 "missing
1": the fields no reference test pins (hyperscore, longest_b / y, poisson, delta_*, average_ppm, tie order, chimera output) had
ONE reading shared by oracle and kernels; this is the second pin.

What it takes from elsewhere: the peptide list (sequence, modifications, monoisotopic, nterm, decoy, missed cleavages) — the
output of `Parameters::build`'s digest, which the reference's own unit tests pin (`enzyme.rs:401-812`, `peptide.rs:429-720`,
`database.rs:595-671`, restated in oracle/selftest.cpp) and which is not part of the scoring arithmetic re-read here.
"""
import math

import numpy as np

f32 = np.float32
H2O, PROTON, NEUTRON = f32(18.010565), f32(1.0072764), f32(1.00335)          # mass.rs:5-7
MONO = np.array([71.03711, 0.0, 103.00919, 115.02694, 129.04259, 147.0684, 57.02146, 137.05891, 113.08406, 0.0, 128.09496,
                 113.08406, 131.0405, 114.04293, 237.14774, 97.05276, 128.05858, 156.1011, 87.03203, 101.04768, 150.95363,
                 99.06841, 186.07932, 0.0, 163.06332, 0.0], dtype=np.float32)  # mass.rs:64-69
U32MAX = 0xFFFFFFFF
A, B, C_, X, Y, Z = range(6)                                                  # ion_series.rs:8-15 (the C ABI's numbering)


def bounds(tol, center):
    """Tolerance::bounds, mass.rs:21-35.  tol = (kind, lo, hi)."""
    kind, lo, hi = tol
    center, lo, hi = f32(center), f32(lo), f32(hi)
    if kind == "ppm":
        return center + center * lo / f32(1_000_000.0), center + center * hi / f32(1_000_000.0)
    if kind == "pct":
        return center + center * lo / f32(100.0), center + center * hi / f32(100.0)
    return center + lo, center + hi


def tol_mul(tol, rhs):
    """impl Mul<f32> for Tolerance, mass.rs:47-57"""
    return (tol[0], f32(tol[1]) * f32(rhs), f32(tol[2]) * f32(rhs))


def binary_search_slice(sorted_f32, low, high):
    """database.rs:549-561 over an ascending f32 array (total_cmp == numeric order for the finite values used here)."""
    left = max(int(np.searchsorted(sorted_f32, low, side="left")) - 1, 0)
    right = left + int(np.searchsorted(sorted_f32[left:], high, side="right"))
    return left, right


def max_fragment_charge(user, precursor_charge):
    """scoring.rs:239-247"""
    return max(min(precursor_charge, (user + 1) if user is not None else precursor_charge), 2)


def lnfact(n):
    """scoring.rs:170-177"""
    if n == 0:
        return 1.0
    n = float(n)
    return n * math.log(n) - n + 0.5 * math.log(n) + 0.5 * math.log(math.pi * 2.0 * n)


def score_type_score(score_type, matched_b, matched_y, summed_b, summed_y):
    """ScoreType::score, scoring.rs:179-201"""
    with np.errstate(all="ignore"):
        if score_type == "SageHyperScore":
            i = float(f32(summed_b) + f32(1.0)) * float(f32(summed_y) + f32(1.0))
            s = (math.log(i) if i > 0 else (float("-inf") if i == 0 else float("nan"))) + lnfact(matched_b) + lnfact(matched_y)
        else:
            s = float(np.log1p(f32(summed_b) + f32(summed_y))) + lnfact(matched_b) + lnfact(matched_y)
    return s if math.isfinite(s) else 255.0


class Run:
    """scoring.rs:771-793"""

    def __init__(self):
        self.start = self.length = self.last = self.longest = 0

    def matched(self, index):
        if self.last == index:
            return
        elif self.start + self.length == index:
            self.length += 1
            self.longest = max(self.longest, self.length)
        else:
            self.start = index
            self.length = 1
            self.longest = max(self.longest, self.length)
        self.last = index


def sift_down(a, k, index):
    """heap.rs:41-60 on a[:k]"""
    while index * 2 + 1 < k:
        smallest = index
        if a[index * 2 + 1] < a[smallest]:
            smallest = index * 2 + 1
        if index * 2 + 2 < k and a[index * 2 + 2] < a[smallest]:
            smallest = index * 2 + 2
        if smallest != index:
            a[smallest], a[index] = a[index], a[smallest]
            index = smallest
        else:
            break


def bounded_min_heapify(a, k):
    """heap.rs:7-29"""
    if len(a) <= k:
        return
    for i in reversed(range(k // 2)):
        sift_down(a, k, i)
    for i in range(k, len(a)):
        if a[i] > a[0]:
            a[i], a[0] = a[0], a[i]
            sift_down(a, k, 0)


def select_most_intense_peak(masses, intensities, center, tol):
    """spectrum.rs:134-159 (offset None)"""
    lo, hi = bounds(tol, center)
    i, j = binary_search_slice(masses, lo, hi)
    best, max_int = None, f32(0.0)
    for idx in range(i, j):
        if masses[idx] >= lo and masses[idx] <= hi and intensities[idx] >= max_int:
            max_int = intensities[idx]
            best = idx
    return best


class Peptides:
    """The fields of `Peptide` the path reads (peptide.rs:12-31), from flat arrays."""

    def __init__(self, arrays):
        self.mono = np.ascontiguousarray(arrays["pep_mono"], np.float32)
        self.seq_off = np.asarray(arrays["seq_off"]).astype(np.int64)
        self.seq = np.asarray(arrays["seq"], np.uint8)
        self.mods = np.asarray(arrays["mods"], np.float32)
        self.nterm = np.nan_to_num(np.asarray(arrays["nterm"], np.float32))   # Option<f32>: NaN = None → unwrap_or_default
        self.decoy = np.asarray(arrays["decoy"], np.uint8)
        self.missed = np.asarray(arrays["missed"], np.uint8)
        self.n = len(self.mono)
        self._series = {}

    def length(self, p):
        return int(self.seq_off[p + 1] - self.seq_off[p])

    def ion_series(self, p, kind):
        """IonSeries::new + Iterator::next, ion_series.rs:36-85 → np.float32[L - 1]"""
        key = (p, kind)
        if key in self._series:
            return self._series[key]
        Cc, O, H, PRO, N = f32(12.0), f32(15.994914), f32(1.007825), f32(1.0072764), f32(14.003074)
        NH3 = N + H * f32(2.0) + PRO
        nterm, mono = self.nterm[p], self.mono[p]
        if kind == A:
            cum = nterm - (Cc + O)
        elif kind == B:
            cum = nterm
        elif kind == C_:
            cum = nterm + NH3
        elif kind == X:
            cum = mono - nterm + (Cc + O - NH3 + N + H)
        elif kind == Y:
            cum = mono - nterm
        else:
            cum = mono - nterm - NH3
        a, b = self.seq_off[p], self.seq_off[p + 1]
        out = np.empty(max(b - a - 1, 0), np.float32)
        for i in range(b - a - 1):
            r = self.seq[a + i]
            step = (MONO[r - 65] if 65 <= r <= 90 else f32(0.0)) + self.mods[a + i]
            cum = cum + step if kind in (A, B, C_) else cum + (-step)
            out[i] = cum
        self._series[key] = out
        return out


class SecondScorer:
    """Scorer (scoring.rs:210-232) over a peptide list; `p` = a sage_amd.api.ScorerParams (field names are the reference's)."""

    def __init__(self, peptides, ion_kinds, min_ion_index, p):
        self.db, self.kinds, self.min_ion_index, self.p = peptides, list(ion_kinds), min_ion_index, p
        self.precursor_tol = (p.precursor_tol.kind, p.precursor_tol.lo, p.precursor_tol.hi)
        self.fragment_tol = (p.fragment_tol.kind, p.fragment_tol.lo, p.fragment_tol.hi)
        self._stored = {}

    # ---- the fragment index's contents, by brute force --------------------------------------------------------------------
    def stored_fragments(self, pep):
        """database.rs:281-292: the ions build_from_peptides keeps for peptide `pep` (fragment_mz values)."""
        if pep not in self._stored:
            L = self.db.length(pep)
            keep = []
            for kind in self.kinds:
                s = self.db.ion_series(pep, kind)
                for ion_idx in range(len(s)):
                    ok = (ion_idx + 1) > self.min_ion_index if kind in (A, B, C_) else (max(L - 1, 0) - ion_idx) > self.min_ion_index
                    if ok:
                        keep.append(s[ion_idx])
            self._stored[pep] = np.array(keep, np.float32)
        return self._stored[pep]

    # ---- scoring.rs:335-382 ---------------------------------------------------------------------------------------------------
    def matched_peaks_with_isotope(self, masses, precursor_mass, precursor_charge, precursor_tol, isotope_error):
        db = self.db
        center = f32(precursor_mass) - f32(isotope_error) * NEUTRON
        plo, phi = bounds(precursor_tol, center)
        pre_lo, pre_hi = binary_search_slice(db.mono, plo, phi)                   # database.rs:410-415
        mfc = max_fragment_charge(self.p.max_fragment_charge, precursor_charge)
        potential = pre_hi - pre_lo + 1
        prelim = [(0, U32MAX, 0, 0)] * potential                                   # PreScore::default()
        matched = np.zeros(potential, np.int64)
        # every Theoretical whose peptide passes page_search's precursor predicate (database.rs:526-531)
        peps, frags = [], []
        for pep in range(pre_lo, min(pre_hi, db.n - 1) + 1):
            if not (pep > pre_lo or (pep == pre_lo and db.mono[pep] >= plo)):
                continue
            if not (pep < pre_hi or (pep == pre_hi and db.mono[pep] <= phi)):
                continue
            s = self.stored_fragments(pep)
            peps.append(np.full(len(s), pep - pre_lo, np.int64))
            frags.append(s)
        total = 0
        if frags:
            peps, frags = np.concatenate(peps), np.concatenate(frags)
            for peak_mass in masses:
                for charge in range(1, mfc):
                    flo, fhi = bounds(self.fragment_tol, f32(peak_mass) * f32(charge))   # scoring.rs:360, database.rs:482
                    hit = (frags >= flo) & (frags <= fhi)                                # database.rs:532-533
                    if hit.any():
                        np.add.at(matched, peps[hit], 1)
                        total += int(hit.sum())
        scored = 0
        for idx in np.flatnonzero(matched):
            assert matched[idx] < 65536
            prelim[idx] = (int(matched[idx]), pre_lo + int(idx), precursor_charge, isotope_error)
            scored += 1
        hits = [total, scored, prelim]
        if total == 0:
            return hits
        self.trim_hits(hits)
        return hits

    def trim_hits(self, hits):
        """scoring.rs:322-329"""
        n = len(hits[2])
        lo, hi = min(self.p.report_psms * 2, n), n
        k = lo if 50 < lo else (hi if 50 > hi else 50)                              # 50.clamp(lo, hi)
        bounded_min_heapify(hits[2], k)
        del hits[2][k:]

    def matched_peaks(self, masses, precursor_mass, precursor_charge, precursor_tol):
        """scoring.rs:384-416"""
        if self.p.min_isotope_err != self.p.max_isotope_err:
            hits = [0, 0, []]
            for isotope in range(self.p.min_isotope_err, self.p.max_isotope_err + 1):
                h = self.matched_peaks_with_isotope(masses, precursor_mass, precursor_charge, precursor_tol, isotope)
                hits[0] += h[0]; hits[1] += h[1]; hits[2].extend(h[2])
            self.trim_hits(hits)
            return hits
        return self.matched_peaks_with_isotope(masses, precursor_mass, precursor_charge, precursor_tol, 0)

    def initial_hits(self, masses, prec_mz, prec_charge, isolation):
        """scoring.rs:418-462.  prec_charge: None / 0 = not annotated; isolation: (lo, hi) Da or None."""
        mz = f32(prec_mz) - PROTON
        p = self.p
        if p.wide_window:
            hits = [0, 0, []]
            for z in range(p.min_precursor_charge, p.max_precursor_charge + 1):
                tol = tol_mul(("da", isolation[0], isolation[1]) if isolation is not None else ("da", -2.4, 2.4), f32(z))
                h = self.matched_peaks(masses, mz * f32(z), z, tol)
                hits[0] += h[0]; hits[1] += h[1]; hits[2].extend(h[2])
        elif prec_charge and not p.override_precursor_charge:
            hits = self.matched_peaks(masses, mz * f32(prec_charge), prec_charge, self.precursor_tol)
        else:
            hits = [0, 0, []]
            for z in range(p.min_precursor_charge, p.max_precursor_charge + 1):
                h = self.matched_peaks(masses, mz * f32(z), z, self.precursor_tol)
                hits[0] += h[0]; hits[1] += h[1]; hits[2].extend(h[2])
        self.trim_hits(hits)
        return hits

    # ---- scoring.rs:675-767 ---------------------------------------------------------------------------------------------------
    def score_candidate(self, masses, intensities, pre):
        _, pep, charge, iso = pre
        s = dict(peptide=pep, precursor_charge=charge, isotope_error=iso, matched_b=0, matched_y=0, summed_b=f32(0), summed_y=f32(0),
                 ppm_difference=f32(0))
        mfc = max_fragment_charge(self.p.max_fragment_charge, charge)
        b_run, y_run = Run(), Run()
        with np.errstate(all="ignore"):
            for kind in self.kinds:
                series = self.db.ion_series(pep, kind)
                for idx in range(len(series)):
                    for z in range(1, mfc):
                        mz = series[idx] / f32(z)
                        peak = select_most_intense_peak(masses, intensities, mz, self.fragment_tol)
                        if peak is None:
                            continue
                        pm, pi = masses[peak], intensities[peak]
                        s["ppm_difference"] = s["ppm_difference"] + pi * abs(mz - pm) * f32(2E6) / (mz + pm)
                        if kind in (A, B, C_):
                            s["matched_b"] += 1
                            s["summed_b"] = s["summed_b"] + pi
                            b_run.matched(idx)
                        else:
                            s["matched_y"] += 1
                            s["summed_y"] = s["summed_y"] + pi
                            y_run.matched(idx)
            s["hyperscore"] = score_type_score(self.p.score_type, s["matched_b"], s["matched_y"], s["summed_b"], s["summed_y"])
            s["longest_b"], s["longest_y"] = b_run.longest, y_run.longest
            s["ppm_difference"] = s["ppm_difference"] / (s["summed_b"] + s["summed_y"])
        return s

    # ---- scoring.rs:478-595 ---------------------------------------------------------------------------------------------------
    def build_features(self, spec, masses, intensities, tic, hits, report_psms, features):
        sv = [self.score_candidate(masses, intensities, pre) for pre in hits[2] if pre[1] != U32MAX]
        sv = [s for s in sv if s["matched_b"] + s["matched_y"] >= self.p.min_matched_peaks]
        sv.sort(key=lambda s: -s["hyperscore"])                                       # stable, descending (scoring.rs:495)
        lam = (hits[0] / hits[1]) if hits[1] else float("nan")
        mz = f32(spec["prec_mz"]) - PROTON
        with np.errstate(all="ignore"):
            for idx in range(min(report_psms, len(sv))):
                s = sv[idx]
                pep = s["peptide"]
                precursor_mass = mz * f32(s["precursor_charge"])
                nxt = sv[idx + 1]["hyperscore"] if idx + 1 < len(sv) else 0.0
                best = sv[0]["hyperscore"]
                k = s["matched_b"] + s["matched_y"]
                try:
                    poisson = (k * math.log(lam) - lam - lnfact(k)) / math.log(10.0)
                except ValueError:
                    poisson = float("nan")
                isotope_error = f32(s["isotope_error"]) * NEUTRON
                mono = self.db.mono[pep]
                delta_mass = (precursor_mass - mono - isotope_error) * f32(2E6) / (precursor_mass - isotope_error + mono)
                L = self.db.length(pep)
                features.append(dict(
                    peptide_idx=pep, rank=idx + 1, label=-1 if self.db.decoy[pep] else 1, expmass=precursor_mass, calcmass=mono,
                    charge=s["precursor_charge"], rt=f32(spec["rt"]), ims=f32(spec["ims"]), delta_mass=delta_mass,
                    isotope_error=isotope_error, average_ppm=s["ppm_difference"], hyperscore=s["hyperscore"],
                    delta_next=s["hyperscore"] - nxt, delta_best=best - s["hyperscore"], matched_peaks=k,
                    matched_intensity_pct=f32(100.0) * (s["summed_b"] + s["summed_y"]) / f32(tic),
                    poisson=poisson if math.isfinite(poisson) else float("-inf"), longest_b=s["longest_b"],
                    longest_y=s["longest_y"], longest_y_pct=f32(s["longest_y"]) / f32(L), peptide_len=L,
                    scored_candidates=hits[1], missed_cleavages=int(self.db.missed[pep]),
                    ms2_intensity=s["summed_b"] + s["summed_y"], file_id=spec["file_id"]))

    # ---- scoring.rs:598-644 ---------------------------------------------------------------------------------------------------
    def remove_matched_peaks(self, masses, intensities, psm):
        mfc = max_fragment_charge(self.p.max_fragment_charge, psm["charge"])
        to_remove = []
        for kind in self.kinds:
            for ion in self.db.ion_series(psm["peptide_idx"], kind):
                for z in range(1, mfc):
                    peak = select_most_intense_peak(masses, intensities, ion / f32(z), self.fragment_tol)
                    if peak is not None:
                        to_remove.append((masses[peak], intensities[peak]))
        keep = [i for i in range(len(masses)) if (masses[i], intensities[i]) not in to_remove]
        m, it = masses[keep], intensities[keep]
        tic = f32(0.0)
        for v in it:
            tic = tic + v
        return m, it, tic

    # ---- scoring.rs:300-309, 465-474, 648-672 ---------------------------------------------------------------------------------
    def score(self, spec):
        """spec: dict(masses, intensities, tic, prec_mz, prec_charge, isolation, rt, ims, file_id) → list of Feature dicts"""
        masses, intens, tic = spec["masses"], spec["intensities"], f32(spec["tic"])
        hits = self.initial_hits(masses, spec["prec_mz"], spec["prec_charge"], spec["isolation"])
        feats = []
        if not self.p.chimera:
            self.build_features(spec, masses, intens, tic, hits, self.p.report_psms, feats)
            return feats, hits
        prev = 0
        while len(feats) < self.p.report_psms:
            self.build_features(spec, masses, intens, tic, hits, 1, feats)
            if len(feats) > prev:
                masses, intens, tic = self.remove_matched_peaks(masses, intens, feats[prev])
                feats[prev]["rank"] = prev + 1
                prev = len(feats)
            else:
                break
        return feats, hits


def spectrum_of(batch, i):
    """Spectrum i of a sage_amd.api.SpectrumBatch as the dict SecondScorer.score takes."""
    a, b = int(batch.peak_off[i]), int(batch.peak_off[i + 1])
    iso = None
    if batch.isolation_lo is not None and not np.isnan(batch.isolation_lo[i]):
        iso = (batch.isolation_lo[i], batch.isolation_hi[i])
    ims = 0.0
    if batch.inverse_ion_mobility is not None and not np.isnan(batch.inverse_ion_mobility[i]):
        ims = batch.inverse_ion_mobility[i]
    return dict(masses=batch.masses[a:b], intensities=batch.intensities[a:b], tic=batch.total_ion_current[i],
                prec_mz=batch.precursor_mz[i], prec_charge=int(batch.precursor_charge[i]), isolation=iso,
                rt=batch.scan_start_time[i] if batch.scan_start_time is not None else 0.0, ims=ims,
                file_id=int(batch.file_id[i]) if batch.file_id is not None else 0)
