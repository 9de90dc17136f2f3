// core.h — arithmetic shared by the HIP kernels (device) and the host-side unit tests.
//
// Everything here is a pure function over plain pointers so that the same source compiles for
// gfx950 (as __device__ code inside kernels.hip) and for the host (tests/hostemu/core_emu.cpp,
// tests/test_core_emulation.py), which lets the order-sensitive f32 arithmetic, the k-select emulation
// and the peak matching be checked without a GPU.  All f32 expressions keep the reference's operation order; translation
// units including this file MUST be compiled with -ffp-contract=off (no FMA contraction).
//
// Reference citations: /root/reference/crates/sage/src/<file>:<line>.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SAGE_HD __host__ __device__ __forceinline__
#else
#define SAGE_HD inline
#endif

namespace sagecore {

constexpr float PROTON = 1.0072764f;  // mass.rs:6
constexpr float NEUTRON = 1.00335f;   // mass.rs:7

struct Tol {  // mass.rs:10-16
    int kind; // 0 ppm, 1 pct, 2 da
    float lo, hi;
};

// Tolerance::bounds, mass.rs:21-35 — (center*lo)/1e6 then center + delta, in f32
SAGE_HD void tol_bounds(const Tol& t, float center, float& out_lo, float& out_hi) {
    if (t.kind == 0) {
        float dlo = center * t.lo / 1000000.0f;
        float dhi = center * t.hi / 1000000.0f;
        out_lo = center + dlo;
        out_hi = center + dhi;
    } else if (t.kind == 1) {
        float dlo = center * t.lo / 100.0f;
        float dhi = center * t.hi / 100.0f;
        out_lo = center + dlo;
        out_hi = center + dhi;
    } else {
        out_lo = center + t.lo;
        out_hi = center + t.hi;
    }
}

// Tolerance::bounds with one division instead of two when a ppm tolerance is symmetric (lo == -hi): center * -h == -(center * h)
// and x / 1e6 is sign-symmetric under round-to-nearest, so the lower delta is exactly the negated upper one (the rescoring
// kernel's per-match call; tests/test_core_emulation.py compares the bits with tol_bounds)
SAGE_HD void tol_bounds_sym(const Tol& t, bool symmetric, float center, float& lo, float& hi) {
    if (symmetric && t.kind == 0) {
        const float d = center * t.hi / 1000000.0f;
        lo = center + -d;
        hi = center + d;
    } else {
        tol_bounds(t, center, lo, hi);
    }
}

// ---- x / 1e6 and x / 3 without the IEEE division sequence (12 vector instructions on gfx950, v_rcp_f32 and the div_scale /
// div_fmas / div_fixup trio among them): q0 = x * RN(1 / c); e = fma(-q0, c, x), the exact remainder; q = fma(e, RN(1 / c), q0).
// For c = 1e6 and c = 3 the result is the correctly rounded quotient — the bits of `x / c` — for EVERY f32 x except +-inf (NaN
// instead of inf), -0 (+0 instead of -0) and, for 1e6, 536 values of |x| <= 0x1.feb14ep-121: tests/hostemu/div_const_proof.c
// walks all 2^32 inputs (tests/test_core_emulation.py runs it).  The callers use it only where the host has shown that every
// dividend lies in FAST_DIV_LO <= |x| <= FAST_DIV_HI (capi.hip: scorer_tol_mode, from the range of the ion table and the
// tolerance), far inside — no zero, no infinity, no denormal quotient, remainder or product on the way.
constexpr float FAST_DIV_LO = 0x1p-60f, FAST_DIV_HI = 0x1p100f;
SAGE_HD float div_const_fast(float x, float c, float rc) {
    const float q0 = x * rc;
    const float e = __builtin_fmaf(-q0, c, x);
    return __builtin_fmaf(e, rc, q0);
}
SAGE_HD float div_1e6_fast(float x) { return div_const_fast(x, 1000000.0f, 1.0f / 1000000.0f); }
SAGE_HD float div_3_fast(float x) { return div_const_fast(x, 3.0f, 1.0f / 3.0f); }

// Tolerance::bounds as the rescoring kernels call it, once per (candidate ion, charge) that passed the bitmap, and the m/z of the
// fragment at that charge.  FAST is a property of the kernel INSTANCE (kernels.hip: rescore_kernel<.., .., true>), chosen by the
// host once per scorer (TOL_FAST of DevScorer::tol_mode): every division by 1e6 and by 3 takes the short form, and the instance
// carries no IEEE sequence for them at all — a run-time choice between the two forms at the four call sites cost the kernel more
// (registers, code) than the short form saves.  `symmetric`: lo == -hi of a ppm tolerance (one division: tol_bounds_sym).
enum TolMode : uint32_t { TOL_SYM = 1u, TOL_FAST = 2u };
template <bool FAST>
SAGE_HD void tol_bounds_mode(const Tol& t, bool symmetric, float center, float& lo, float& hi) {
    if (FAST && t.kind == 0) {
        if (symmetric) {
            const float d = div_1e6_fast(center * t.hi);
            lo = center + -d;
            hi = center + d;
        } else {
            lo = center + div_1e6_fast(center * t.lo);
            hi = center + div_1e6_fast(center * t.hi);
        }
    } else {
        tol_bounds_sym(t, symmetric, center, lo, hi);
    }
}
// theoretical m/z of a fragment at charge c: monoisotopic_mass / charge as f32 (scoring.rs:707) — x / 1 and x / 2 are x and
// x * 0.5 bit for bit
template <bool FAST>
SAGE_HD float fragment_mz(float ion, uint32_t c) {
    if (c == 1) return ion;
    if (c == 2) return ion * 0.5f;
    if (FAST && c == 3) return div_3_fast(ion);
    return ion / (float)c;
}

SAGE_HD Tol tol_scaled(const Tol& t, float rhs) {  // impl Mul<f32>, mass.rs:47-57
    Tol r;
    r.kind = t.kind;
    r.lo = t.lo * rhs;
    r.hi = t.hi * rhs;
    return r;
}

// f32::total_cmp as an integer key: total_cmp(a,b) == compare(order_key(a), order_key(b))
SAGE_HD int32_t order_key(float f) {
    union { float f; int32_t i; } u;
    u.f = f;
    return u.i ^ (int32_t)(((uint32_t)(u.i >> 31)) >> 1);
}
SAGE_HD int64_t order_key64(double d) {
    union { double d; int64_t i; } u;
    u.d = d;
    return u.i ^ (int64_t)(((uint64_t)(u.i >> 63)) >> 1);
}

// scoring.rs:239-247 — exclusive upper bound of the fragment charge loop; user < 0 == None
SAGE_HD uint32_t max_fragment_charge(int user, uint32_t precursor_charge) {
    uint32_t inner = user >= 0 ? (uint32_t)((user + 1) & 0xFF) : precursor_charge;
    uint32_t m = precursor_charge < inner ? precursor_charge : inner;
    return m < 2 ? 2 : m;
}

// ---- PreScore (scoring.rs:43-49) packed so that u64 compare == derived lexicographic Ord ------
constexpr uint64_t PRESCORE_EMPTY = 0x0000FFFFFFFF0080ull;  // (0, u32::MAX, 0, 0)
SAGE_HD uint64_t pack_prescore(uint32_t matched, uint32_t peptide, uint32_t charge, int iso) {
    return ((uint64_t)(matched & 0xFFFF) << 48) | ((uint64_t)peptide << 16) | ((uint64_t)(charge & 0xFF) << 8) |
           (uint64_t)((iso + 128) & 0xFF);
}
SAGE_HD uint32_t prescore_matched(uint64_t p) { return (uint32_t)(p >> 48); }
SAGE_HD uint32_t prescore_peptide(uint64_t p) { return (uint32_t)((p >> 16) & 0xFFFFFFFFu); }
SAGE_HD uint32_t prescore_charge(uint64_t p) { return (uint32_t)((p >> 8) & 0xFF); }
SAGE_HD int prescore_iso(uint64_t p) { return (int)(p & 0xFF) - 128; }

// trim_hits' k (scoring.rs:323-326): 50.clamp(min(2*report_psms, len), len)
SAGE_HD uint32_t trim_k(uint64_t len, uint32_t report_psms) {
    uint64_t lo = (uint64_t)report_psms * 2 < len ? (uint64_t)report_psms * 2 : len;
    uint64_t k = 50 > lo ? 50 : lo;
    return (uint32_t)(k < len ? k : len);
}

// heap.rs:40-60
SAGE_HD void sift_down(uint64_t* h, uint32_t len, uint32_t index) {
    for (;;) {
        uint32_t l = index * 2 + 1;
        if (l >= len) break;
        uint32_t smallest = index;
        if (h[l] < h[smallest]) smallest = l;
        uint32_t r = l + 1;
        if (r < len && h[r] < h[smallest]) smallest = r;
        if (smallest == index) break;
        uint64_t t = h[smallest];
        h[smallest] = h[index];
        h[index] = t;
        index = smallest;
    }
}
// first loop of bounded_min_heapify (heap.rs:13-15)
SAGE_HD void heap_build(uint64_t* h, uint32_t k) {
    for (uint32_t i = k / 2; i-- > 0;) sift_down(h, k, i);
}
// one step of the scan loop (heap.rs:21-27)
SAGE_HD void heap_offer(uint64_t* h, uint32_t k, uint64_t v) {
    if (k && v > h[0]) {
        h[0] = v;  // slice.swap(i, 0): the displaced minimum lands beyond k and is truncated away
        sift_down(h, k, 0);
    }
}

// ---- CList: a compact stand-in for InitialHits.preliminary (scoring.rs:57) -------------------
// The reference materialises every candidate slot of every precursor-window query, most of them
// PreScore::default().  Only two things about that vector are observable downstream: its first
// trim_k() entries verbatim (they seed the heap) and the order of the later non-empty entries
// (empties can never displace a heap minimum, and are filtered at scoring.rs:489).  A CList keeps
// exactly that: the first `kmax` logical entries verbatim, then non-empty entries only, plus the
// logical length.  kmax = max(50, 2*report_psms) bounds every trim_k().
struct CList {
    uint64_t* items;
    uint32_t stored;
    uint32_t cap;
    uint64_t len;  // logical Vec length
};
SAGE_HD void clist_clear(CList& c) { c.stored = 0; c.len = 0; }
SAGE_HD bool clist_push(CList& c, uint64_t v, uint32_t kmax) {
    if (c.len < kmax || v != PRESCORE_EMPTY) {
        if (c.stored >= c.cap) return false;
        c.items[c.stored++] = v;
    }
    c.len++;
    return true;
}
SAGE_HD bool clist_push_empties(CList& c, uint64_t n, uint32_t kmax) {
    while (n && c.len < kmax) {
        if (c.stored >= c.cap) return false;
        c.items[c.stored++] = PRESCORE_EMPTY;
        c.len++;
        n--;
    }
    c.len += n;
    return true;
}
// trim_hits (scoring.rs:322-329) on a CList
SAGE_HD void clist_trim(CList& c, uint32_t report_psms) {
    uint32_t k = trim_k(c.len, report_psms);
    if (c.len > k) {  // bounded_min_heapify(slice, k), heap.rs:7-28
        heap_build(c.items, k);
        for (uint32_t i = k; i < c.stored; i++) heap_offer(c.items, k, c.items[i]);
    }
    c.stored = k;  // truncate(k)
    c.len = k;
}

// ---- fragment matching -------------------------------------------------------------------------
// number of entries of sorted[0..n) that are <= v  (partition_point(|x| x <= v))
SAGE_HD uint32_t count_le(const float* sorted, uint32_t n, float v) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (sorted[mid] <= v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// number of entries < v
SAGE_HD uint32_t count_lt(const float* sorted, uint32_t n, float v) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (sorted[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// How many experimental peaks i have frag in [win_lo[i], win_hi[i]]?  (the predicate at
// database.rs:532-533 with the window built at database.rs:481 from scoring.rs:360.)  Requires both
// arrays ascending and win_lo[i] <= win_hi[i]; the caller checks that and otherwise uses the scan.
SAGE_HD uint32_t count_windows_sorted(const float* win_lo, const float* win_hi, uint32_t n, float frag) {
    return count_le(win_lo, n, frag) - count_lt(win_hi, n, frag);
}
SAGE_HD uint32_t count_windows_scan(const float* win_lo, const float* win_hi, uint32_t n, float frag) {
    uint32_t c = 0;
    for (uint32_t i = 0; i < n; i++) c += (frag >= win_lo[i] && frag <= win_hi[i]) ? 1u : 0u;
    return c;
}

// Lock-step variant of count_windows_sorted over NFZ fragment charges: all 2*NFZ binary searches
// advance together through fixed power-of-two steps (no data-dependent branches), so their LDS loads
// overlap instead of forming 2*NFZ serial chains.  `top` = largest power of two <= n (0 when n == 0).
template <int NFZ>
SAGE_HD uint32_t count_windows_lockstep(const float* win_lo, const float* win_hi, uint32_t stride, uint32_t n,
                                        uint32_t top, float frag) {
    uint32_t a[NFZ], b[NFZ];
#pragma unroll
    for (int z = 0; z < NFZ; z++) a[z] = b[z] = 0;
    for (uint32_t step = top; step; step >>= 1) {
#pragma unroll
        for (int z = 0; z < NFZ; z++) {
            const uint32_t ia = a[z] + step, ib = b[z] + step;
            const float va = win_lo[(uint32_t)z * stride + (ia < n ? ia : n) - 1];
            const float vb = win_hi[(uint32_t)z * stride + (ib < n ? ib : n) - 1];
            a[z] = (ia <= n && va <= frag) ? ia : a[z];   // a = #{lo <= frag}
            b[z] = (ib <= n && vb < frag) ? ib : b[z];    // b = #{hi <  frag}
        }
    }
    uint32_t c = 0;
#pragma unroll
    for (int z = 0; z < NFZ; z++) c += a[z] - b[z];
    return c;
}
SAGE_HD uint32_t pow2_floor(uint32_t n) {
    uint32_t t = 0;
    if (n) { t = 1; while ((t << 1) <= n && (t << 1)) t <<= 1; }
    return t;
}

// ---- select_most_intense_peak (spectrum.rs:134-159) with offset == None ------------------------
// binary_search_slice(masses, total_cmp, lo, hi) (database.rs:549-561) followed by the filtered scan.
SAGE_HD int select_most_intense_peak(const float* masses, const float* intensities, uint32_t n, float center,
                                     const Tol& tol) {
    float lo, hi;
    tol_bounds(tol, center, lo, hi);
    const int32_t klo = order_key(lo), khi = order_key(hi);
    uint32_t a = 0, b = n;
    while (a < b) {  // partition_point(mass.total_cmp(lo) == Less)
        uint32_t mid = (a + b) >> 1;
        if (order_key(masses[mid]) < klo) a = mid + 1; else b = mid;
    }
    const uint32_t left = a ? a - 1 : 0;
    a = left; b = n;
    while (a < b) {  // partition_point(mass.total_cmp(hi) != Greater)
        uint32_t mid = (a + b) >> 1;
        if (order_key(masses[mid]) <= khi) a = mid + 1; else b = mid;
    }
    const uint32_t right = a;
    int best = -1;
    float max_int = 0.0f;
    for (uint32_t idx = left; idx < right; idx++) {
        const float m = masses[idx];
        if (m >= lo && m <= hi) {
            const float it = intensities[idx];
            if (it >= max_int) {
                max_int = it;
                best = (int)idx;
            }
        }
    }
    return best;
}

// Same result as select_most_intense_peak, with the two partition points found in lock step
// (`top` = pow2_floor(n)).  right = left + partition_point(slice[left..], <= hi) == max(left, #{m <= hi}).
SAGE_HD int select_most_intense_peak_lockstep(const float* masses, const float* intensities, uint32_t n, uint32_t top,
                                              float center, const Tol& tol) {
    float lo, hi;
    tol_bounds(tol, center, lo, hi);
    const int32_t klo = order_key(lo), khi = order_key(hi);
    uint32_t a = 0, b = 0;
    for (uint32_t step = top; step; step >>= 1) {
        const uint32_t ia = a + step, ib = b + step;
        const int32_t ka = order_key(masses[(ia < n ? ia : n) - 1]);
        const int32_t kb = order_key(masses[(ib < n ? ib : n) - 1]);
        a = (ia <= n && ka < klo) ? ia : a;
        b = (ib <= n && kb <= khi) ? ib : b;
    }
    const uint32_t left = a ? a - 1 : 0;
    const uint32_t right = b > left ? b : left;
    int best = -1;
    float max_int = 0.0f;
    for (uint32_t idx = left; idx < right; idx++) {
        const float m = masses[idx];
        if (m >= lo && m <= hi) {
            const float it = intensities[idx];
            if (it >= max_int) {
                max_int = it;
                best = (int)idx;
            }
        }
    }
    return best;
}

// ---- Run (scoring.rs:771-793) -------------------------------------------------------------------
struct Run {
    uint32_t start, length, last, longest;
};
SAGE_HD void run_matched(Run& r, uint32_t index) {
    if (r.last == index) return;
    if (r.start + r.length == index) {
        r.length += 1;
        if (r.length > r.longest) r.longest = r.length;
    } else {
        r.start = index;
        r.length = 1;
        if (r.length > r.longest) r.longest = r.length;
    }
    r.last = index;
}

// Run in ONE register (the rescoring kernel is short of them).  After any update `last == index` and `start + length ==
// index + 1`, so (next = start + length, length, longest) is the whole state — `last` is next - 1, or the initial 0 while next
// is still 0 (the reference's quirk that a first match at index 0 is ignored is kept) — 10 bits each: ion indices stay below
// 1023 (capi.hip refuses longer peptides).  tests/test_core_emulation.py holds it to run_matched.
SAGE_HD void run_matched_packed(uint32_t& r, uint32_t index) {
    const uint32_t next = r & 1023u, length = (r >> 10) & 1023u, longest = r >> 20;
    if ((next ? next - 1u : 0u) == index) return;  // self.last == index
    const uint32_t nl = next == index ? length + 1u : 1u;
    r = (index + 1u) | (nl << 10) | ((nl > longest ? nl : longest) << 20);
}
SAGE_HD uint32_t run_longest_packed(uint32_t r) { return r >> 20; }
// ... and in TWO registers, 21 bits per field, for databases with peptides of more than 1023 residues (the instance of the
// rescoring kernel such a database is scored with; peptide lengths are 16-bit in the device records): same update, wider fields.
SAGE_HD void run_matched_packed(uint64_t& r, uint32_t index) {
    const uint32_t next = (uint32_t)(r & 0x1FFFFFu), length = (uint32_t)((r >> 21) & 0x1FFFFFu), longest = (uint32_t)(r >> 42);
    if ((next ? next - 1u : 0u) == index) return;  // self.last == index
    const uint32_t nl = next == index ? length + 1u : 1u;
    r = (uint64_t)(index + 1u) | ((uint64_t)nl << 21) | ((uint64_t)(nl > longest ? nl : longest) << 42);
}
SAGE_HD uint32_t run_longest_packed(uint64_t r) { return (uint32_t)(r >> 42); }

// ---- Score (scoring.rs:17-30) -------------------------------------------------------------------
struct Score {
    uint32_t peptide;
    uint32_t matched_b, matched_y;  // u16 in the reference
    float summed_b, summed_y;
    uint32_t longest_b, longest_y;
    float ppm_difference;
    uint32_t precursor_charge;
    int isotope_error;
};

// score_candidate's accumulation loop (scoring.rs:699-759) over a peptide's precomputed ion table:
// ions[k*(L-1) + idx] is IonSeries(peptide, ion_kinds[k]).nth(idx) (ion_series.rs:68-85).
SAGE_HD void score_candidate(Score& s, const float* ions, uint32_t lm1, const uint8_t* ion_kinds, uint32_t n_kinds,
                             uint32_t max_fc, const float* masses, const float* intensities, uint32_t n_peaks,
                             const Tol& fragment_tol) {
    Run b_run = {0, 0, 0, 0}, y_run = {0, 0, 0, 0};
    s.matched_b = s.matched_y = 0;
    s.summed_b = s.summed_y = 0.0f;
    s.ppm_difference = 0.0f;
    for (uint32_t k = 0; k < n_kinds; k++) {
        const bool nterm_kind = ion_kinds[k] <= 2;  // A | B | C
        const float* series = ions + (uint64_t)k * lm1;
        for (uint32_t idx = 0; idx < lm1; idx++) {
            const float frag = series[idx];
            for (uint32_t charge = 1; charge < max_fc; charge++) {
                const float mz = frag / (float)charge;
                const int pk = select_most_intense_peak(masses, intensities, n_peaks, mz, fragment_tol);
                if (pk < 0) continue;
                const float peak_mass = masses[pk];
                const float peak_intensity = intensities[pk];
                const float d = __builtin_fabsf(mz - peak_mass);
                s.ppm_difference += peak_intensity * d * 2E6f / (mz + peak_mass);
                if (nterm_kind) {
                    s.matched_b += 1;
                    s.summed_b += peak_intensity;
                    run_matched(b_run, idx);
                } else {
                    s.matched_y += 1;
                    s.summed_y += peak_intensity;
                    run_matched(y_run, idx);
                }
            }
        }
    }
    s.longest_b = b_run.longest;
    s.longest_y = y_run.longest;
    s.ppm_difference /= s.summed_b + s.summed_y;
}

// ---- position tables of the tile-major index copies ------------------------------------------------------------------------
// lut[t][c] = first position of tile t whose m/z is >= c / scale; scale is a power of two, so lo * scale, hi * scale and
// c / scale are exact and a window [lo, hi] needs NO safety margin: its entries start at cell floor(lo * scale) and end
// before cell floor(hi * scale) + 1.  Row entry 0 is the tile's start and the last one (stride - 1) its end, whatever the
// m/z there (NaN and m/z beyond the table stay in the last cell's run).  tests/test_core_emulation.py checks the pair
// (lut_entry, lut_cells) against a plain scan.
SAGE_HD void lut_cells(float lo, float hi, float scale, uint32_t stride, uint32_t& icl, uint32_t& ich) {
    float cl = __builtin_floorf(lo * scale);
    float ch = __builtin_floorf(hi * scale) + 1.0f;
    cl = cl > 0.0f ? cl : 0.0f;  // also maps NaN to 0
    ch = ch > 0.0f ? ch : 0.0f;
    // a window beyond the table starts at the last real cell (stride - 2), whose run also holds every entry beyond the table
    // (a table capped below the largest m/z); it ends with the tile
    icl = cl < (float)(stride - 2) ? (uint32_t)cl : stride - 2;
    ich = ch < (float)(stride - 1) ? (uint32_t)ch : stride - 1;
    if (!(lo <= hi)) icl = ich = 0;  // empty run
}
// entry c of the row of a tile that spans positions [begin, end) of `mz` (read with a stride of `step` floats)
SAGE_HD uint32_t lut_entry(const float* mz, uint32_t step, uint64_t begin, uint64_t end, uint32_t c, uint32_t stride, float scale) {
    if (c == 0) return (uint32_t)begin;
    if (c == stride - 1) return (uint32_t)end;
    const double edge = (double)c / (double)scale;
    uint64_t lo = begin, hi = end;
    while (lo < hi) {  // partition_point(m/z < edge); NaN compares false and stays at the end
        const uint64_t mid = (lo + hi) >> 1;
        if ((double)mz[mid * step] < edge) lo = mid + 1; else hi = mid;
    }
    return (uint32_t)lo;
}

// ---- the small tiles' position table in succinct form (round 6) ---------------------------------------------------------------
// A row of the table above answers lut[c] = first position of the tile whose m/z is >= c / scale.  For the narrow kernel's small
// tiles most cells are EMPTY (a tile of 2 048 peptides holds ~60 000 entries in ~160 000 cells, and they cluster in mass-defect
// bands) and most (peak, charge) windows find nothing — yet every lookup cost a 128-byte line of a 640 KB row.  The same function
// from two levels: per 32 cells one word of occupancy bits (bit b: cell 32 w + b holds an entry) and the RANK of the word — the
// number of non-empty cells before it, as an index into `pos`, the tile's run starts in cell order with the tile's end as the
// last entry.  Then
//     lut[c] == pos[rank(c)],   rank(c) = l1[c >> 5].rank + popcount(l1[c >> 5].bits & ((1 << (c & 31)) - 1))
// (the first non-empty cell at or after c starts where lut[c] points; no such cell: the tile's end), so a window's run is
// [pos[rank(icl)], pos[rank(ich)]) — and rank(icl) == rank(ich) says "empty" without touching `pos` at all.  The first level is
// 40 KB per tile (it stays in the caches while the tile's spectra are scored), `pos` a third of the old row.  Exactly the old
// function: tests/test_core_emulation.py builds both from random entries and compares every cell.
struct LutWord {
    uint32_t bits, rank;
};
SAGE_HD uint32_t lut_rank(const LutWord& w, uint32_t c) {
    const uint32_t below = w.bits & ((1u << (c & 31u)) - 1u);
#if defined(__HIP_DEVICE_COMPILE__)
    return w.rank + (uint32_t)__popc(below);
#else
    return w.rank + (uint32_t)__builtin_popcount(below);
#endif
}

// ---- select_most_intense_peak through a direct-index table (rescore_kernel) ------------------------------------------------
// plut[b] = number of peaks with mass < b * W (total order).  W is a power of two, so bin(lo) = floor(lo / W) and b * W are
// exact and plut[bin(lo)] <= partition_point(mass < lo): a short forward walk finishes the job.  Same peaks considered and
// the same filtered scan as select_most_intense_peak above.
constexpr uint32_t PLUT_BINS = 256;
SAGE_HD float peak_lut_width(float top) {  // a power of two w >= 1 with PLUT_BINS * w > the largest mass (any power of two is CORRECT:
    // masses beyond the table share its last bin; this one keeps the walks of select_peak_lut short)
    // top = 2^e f, 1 <= f < 2: 2^(e - 7) > top / 256 >= 2^(e - 8).  From the exponent field, no loop (wave-uniform float work runs
    // on the vector ALU all the same); capped at 2^100; a negative, zero or tiny top gives 1, a NaN the cap.
    uint32_t bits;
    __builtin_memcpy(&bits, &top, 4);
    const uint32_t e = (bits >> 23) & 0xFFu;  // biased
    uint32_t we = (bits >> 31) || e < 134u ? 127u : e - 7u;
    we = we > 227u ? 227u : we;
    const uint32_t wb = we << 23;
    float w;
    __builtin_memcpy(&w, &wb, 4);
    return w;
}
SAGE_HD float pow2_reciprocal(float w) {  // exact 1 / w for a normal power of two (peak_lut_width's)
    uint32_t bits;
    __builtin_memcpy(&bits, &w, 4);
    bits = 0x7F000000u - bits;
    float r;
    __builtin_memcpy(&r, &bits, 4);
    return r;
}
SAGE_HD uint32_t peak_lut_entry(const float* pm, uint32_t P, uint32_t b, float w) {
    const int32_t edge = order_key((float)b * w);
    uint32_t lo = 0, hi = P;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (order_key(pm[mid]) < edge) lo = mid + 1; else hi = mid;
    }
    return lo;
}
SAGE_HD int select_peak_lut(const float* pm, const float* pi, uint32_t P, const uint32_t* plut, float inv_w, float lo, float hi) {
    float fb = __builtin_floorf(lo * inv_w);
    fb = fb > 0.0f ? fb : 0.0f;  // also maps NaN to 0
    const uint32_t bin = fb < (float)(PLUT_BINS - 1) ? (uint32_t)fb : PLUT_BINS - 1;
    // Every peak with lo <= mass <= hi lies at or after plut[bin] (all earlier masses are < bin * W <= lo), and the
    // reference's scan over [left, right) keeps exactly those peaks (spectrum.rs:147-157: `mass >= lo && mass <= hi`,
    // most intense wins, the last one on ties).  Walk forward two peaks at a time — their LDS reads are independent —
    // until a mass exceeds hi (masses ascend); intensities are only read for peaks inside the window.
    int best = -1;
    float max_int = 0.0f;
    for (uint32_t a = plut[bin]; a < P; a += 2) {
        const float m0 = pm[a], m1 = pm[a + 1 < P ? a + 1 : a];
        if (m0 >= lo && m0 <= hi) {
            const float it = pi[a];
            if (it >= max_int) { max_int = it; best = (int)a; }
        }
        if (a + 1 < P && m1 >= lo && m1 <= hi) {
            const float it = pi[a + 1];
            if (it >= max_int) { max_int = it; best = (int)(a + 1); }
        }
        if (m0 > hi || m1 > hi || !(hi == hi)) break;
    }
    return best;
}

// ---- 65-ary partition point (kernels.hip: wave_partition_point) --------------------------------------------------------------
// One round of the wavefront-wide search over [lo, hi): 64 pivots at lo + (j + 1) * step - 1, c of them compare true (a prefix);
// the answer then lies in [lo + c * step, min(lo + (c + 1) * step - 1, hi)].  65 * step - 1 >= span, so that the piece behind
// the last pivot reaches hi.  Shared with the host emulation (tests/hostemu/core_emu.cpp: every span and answer up to 400).
SAGE_HD uint32_t wpp_step(uint32_t span) { return span / 65u + 1u; }

// ---- peak-presence bitmap (rescore_kernel's filter in front of select_most_intense_peak) --------------------------------
// PBM_BITS mass bins of a FIXED width of 1/16 Da, the bin index taken modulo PBM_BITS (masses 2048 Da apart share a bin: a
// Bloom filter with one hash).  Every peak sets the bins that overlap [mass - D, mass + D], D bounding |mz - mass| over every
// m/z whose tolerance window (Tolerance::bounds, mass.rs:21-35) contains the peak, plus the roundings; an ion whose bin is clear
// cannot match any peak.  The bin of an ion's m/z at fragment charge c is the bin of the EXACT quotient: x = floor(8 ion) (the
// product by a power of two is exact), floor(x / c) == floor(8 ion / c).  Round 3 sized the bins per spectrum (a power of two
// covering the spectrum's span): that cost a multiply by a per-spectrum scale, a clamp of the conversion (v_med3_f32) and a
// clamp per charge; with the fixed width and the modulo none of them is needed — the conversion saturates on its own, and
// wherever that could matter (negative, non-finite or absurd masses) the bitmap is all ones.
// Conservative by construction (tests/test_core_emulation.py checks it against Tolerance::bounds on adversarial inputs); when
// that cannot be guaranteed (non-finite or negative masses, non-finite tolerances, a relative tolerance of a quarter and more,
// D above 4 Da) every bin is set.
#ifndef SAGE_PBM_LOG2_BITS
#define SAGE_PBM_LOG2_BITS 15  // 32 768 bins (4 KB of LDS; round 4 before: 14 with 1/8 Da bins, 2 % slower — DESIGN.md 4.3)
#endif
#ifndef SAGE_PBM_LOG2_INV_W
#define SAGE_PBM_LOG2_INV_W 4  // bins of 1/16 Da
#endif
constexpr uint32_t PBM_BITS = 1u << SAGE_PBM_LOG2_BITS, PBM_WORDS = PBM_BITS / 32;
constexpr float PBM_INV_W = (float)(1u << SAGE_PBM_LOG2_INV_W);  // bins per Da (a power of two: mass * PBM_INV_W is exact)
constexpr float PBM_MAX_D = 4.0f;  // 64 bins either side
// x of an ion: floor(PBM_INV_W ion), saturating (a negative or NaN ion gives 0, an absurd one 2^32 - 1: such ions match no peak
// of a spectrum the filter is active for).  On the device the conversion is v_cvt_u32_f32 BY NAME: a C++ cast of an out-of-range
// float is undefined (poison to the optimiser), the instruction saturates by definition.
SAGE_HD uint32_t pbm_index(float ion) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float f = ion * PBM_INV_W;
    uint32_t x;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(x) : "v"(f));
    return x;
#else
    const float f = ion * PBM_INV_W;
    return f >= 4294967296.0f ? 0xFFFFFFFFu : f > 0.0f ? (uint32_t)f : 0u;  // (also maps NaN to 0)
#endif
}
SAGE_HD uint32_t pbm_bin_c1(uint32_t x) { return x & (PBM_BITS - 1u); }
SAGE_HD uint32_t pbm_bin_c2(uint32_t x) { return (x >> 1) & (PBM_BITS - 1u); }
SAGE_HD uint32_t pbm_bin_c3(uint32_t x) { return (uint32_t)(((uint64_t)x * 0xAAAAAAABull) >> 33) & (PBM_BITS - 1u); }  // x / 3 for every u32
// D of a peak of mass m is fma(m, a, b): the two coefficients depend on the tolerance alone and are worked out once per scorer
// (the kernel spends ONE instruction per peak on D — every instruction of the one-wavefront-per-spectrum kernels is paid once
// per spectrum whether 1 or 64 lanes need it; round 4: the per-peak form with its division cost ~300 instructions per
// spectrum).  a < 0: no safe bound for this tolerance (every bin of the bitmap is set).
struct PbmReach {
    float a, b;
};
SAGE_HD PbmReach pbm_reach_of(const Tol& t) {
    const float tmax = __builtin_fmaxf(__builtin_fabsf(t.lo), __builtin_fabsf(t.hi));
    const float rel = t.kind == 0 ? tmax * 1.0e-6f : t.kind == 1 ? tmax * 1.0e-2f : 0.0f;
    PbmReach r{-1.0f, 0.0f};
    if (!(tmax == tmax) || !(tmax < 1.0e30f) || !(rel < 0.25f)) return r;
    // relative tolerances: the window of centre c contains m only if |c - m| <= m rel / (1 - rel); 2e-4 relative for the roundings
    // inside Tolerance::bounds and of these coefficients, 2^-20 m for those of ion / charge and of m -+ D, 2^-20 absolute for
    // tiny masses
    r.a = (rel / (1.0f - rel)) * 1.0002f + (1.0f / 1048576.0f);
    r.b = (t.kind == 2 ? tmax * 1.0002f : 0.0f) + (1.0f / 1048576.0f);
    return r;
}
// D of one peak; false: no safe bound (a negative, non-finite or absurd mass, D above PBM_MAX_D)
SAGE_HD bool pbm_peak_reach(const PbmReach& r, float m, float& D) {
    D = __builtin_fmaf(m, r.a, r.b);
    return r.a >= 0.0f && m >= 0.0f && m < 8388608.0f && D <= PBM_MAX_D;
}
// bins [b0, b1] (before the modulo) a peak of mass m >= 0 sets
SAGE_HD void pbm_peak_span(float m, float D, uint32_t& b0, uint32_t& b1) {
    const float f0 = (m - D) * PBM_INV_W, f1 = (m + D) * PBM_INV_W;
    b0 = f0 > 0.0f ? (uint32_t)f0 : 0u;
    b1 = f1 > 0.0f ? (uint32_t)f1 : 0u;
}

}  // namespace sagecore
