// Host build of sage_amd/csrc/core.h — the arithmetic the HIP kernels share with the host — behind a tiny C ABI so that the
// CPU test-suite can exercise it without a GPU (tests/test_core_emulation.py).  TEST INFRASTRUCTURE.
#include <cstdint>
#include <vector>

#include "../../sage_amd/csrc/core.h"

using namespace sagecore;

extern "C" {

void emu_tol_bounds(int kind, float tlo, float thi, float center, float* lo, float* hi) {
    Tol t{kind, tlo, thi};
    tol_bounds(t, center, *lo, *hi);
}
void emu_tol_bounds_sym(float thi, float center, float* lo, float* hi) {  // the kernel's shortcut for ppm(-h, h)
    Tol t{0, -thi, thi};
    tol_bounds_sym(t, t.lo == -t.hi, center, *lo, *hi);
}
// the rescoring kernels' forms (core.h: the FAST instance's short divisions)
void emu_tol_bounds_mode(int kind, float tlo, float thi, uint32_t fast, float center, float* lo, float* hi) {
    Tol t{kind, tlo, thi};
    if (fast) tol_bounds_mode<true>(t, kind == 0 && tlo == -thi, center, *lo, *hi);
    else tol_bounds_mode<false>(t, kind == 0 && tlo == -thi, center, *lo, *hi);
}
float emu_fragment_mz(float ion, uint32_t c, uint32_t fast) { return fast ? fragment_mz<true>(ion, c) : fragment_mz<false>(ion, c); }
float emu_fast_div_lo() { return FAST_DIV_LO; }
float emu_fast_div_hi() { return FAST_DIV_HI; }
uint32_t emu_trim_k(uint64_t len, uint32_t report_psms) { return trim_k(len, report_psms); }
uint32_t emu_max_fragment_charge(int user, uint32_t z) { return max_fragment_charge(user, z); }
int32_t emu_order_key(float f) { return order_key(f); }

// bounded_min_heapify(slice, k) + truncate on packed PreScores (heap.rs:7-28), through CList
uint32_t emu_clist_trim(uint64_t* items, uint32_t n, uint32_t kmax, uint32_t report_psms) {
    std::vector<uint64_t> buf(n + 8);
    CList c{buf.data(), 0, (uint32_t)buf.size(), 0};
    for (uint32_t i = 0; i < n; i++) clist_push(c, items[i], kmax);
    clist_trim(c, report_psms);
    for (uint32_t i = 0; i < c.stored; i++) items[i] = c.items[i];
    return c.stored;
}

uint32_t emu_count_windows(const float* lo, const float* hi, uint32_t n, float frag, int variant) {
    if (variant == 0) return count_windows_scan(lo, hi, n, frag);
    if (variant == 1) return count_windows_sorted(lo, hi, n, frag);
    return count_windows_lockstep<1>(lo, hi, n, n, pow2_floor(n), frag);
}

int emu_select_peak(const float* masses, const float* intens, uint32_t n, float center, int kind, float tlo, float thi, int lockstep) {
    Tol t{kind, tlo, thi};
    return lockstep ? select_most_intense_peak_lockstep(masses, intens, n, pow2_floor(n), center, t)
                    : select_most_intense_peak(masses, intens, n, center, t);
}

// score_candidate's accumulation loop (scoring.rs:699-759) on a precomputed ion table
void emu_score_candidate(const float* ions, uint32_t lm1, const uint8_t* kinds, uint32_t n_kinds, uint32_t max_fc,
                         const float* masses, const float* intens, uint32_t n_peaks, int kind, float tlo, float thi,
                         uint32_t* out_u32 /*matched_b, matched_y, longest_b, longest_y*/, float* out_f32 /*summed_b, summed_y, ppm*/) {
    Tol t{kind, tlo, thi};
    Score s{};
    score_candidate(s, ions, lm1, kinds, n_kinds, max_fc, masses, intens, n_peaks, t);
    out_u32[0] = s.matched_b; out_u32[1] = s.matched_y; out_u32[2] = s.longest_b; out_u32[3] = s.longest_y;
    out_f32[0] = s.summed_b; out_f32[1] = s.summed_y; out_f32[2] = s.ppm_difference;
}

// position table of one tile (core.h: lut_entry) and the runs m windows [lo, hi] read through it (lut_cells): returns the number
// of entries with lo <= m/z <= hi that their run [row[icl], row[ich]) MISSES (must be 0), and the runs' total length
uint64_t emu_lut_windows(const float* mz, uint32_t n, float scale, uint32_t stride, const float* lo, const float* hi, uint32_t m,
                         uint64_t* run_len) {
    std::vector<uint32_t> row(stride);
    for (uint32_t c = 0; c < stride; c++) row[c] = lut_entry(mz, 1, 0, n, c, stride, scale);
    uint64_t missed = 0;
    *run_len = 0;
    for (uint32_t w = 0; w < m; w++) {
        uint32_t icl, ich;
        lut_cells(lo[w], hi[w], scale, stride, icl, ich);
        const uint32_t p0 = row[icl], p1 = row[ich];
        *run_len += p1 > p0 ? p1 - p0 : 0;
        // entries inside the window: mz is sorted up to its NaN-free prefix, so a scan between two binary-search bounds suffices
        for (uint32_t i = 0; i < n; i++)
            if (mz[i] >= lo[w] && mz[i] <= hi[w] && !(i >= p0 && i < p1)) missed++;
    }
    return missed;
}

// wave_partition_point (kernels.hip) with the 64 lanes as a loop: the rounds' arithmetic is core.h's (wpp_step), the pivots /
// ballot / narrowing as the kernel does them.  partition_point over sorted a[lo..hi) of key(a[i]) < bound (strict) or <= bound.
uint32_t emu_wave_partition_point(const float* a, uint32_t lo, uint32_t hi, float bound_value, int strict) {
    const int32_t bound = order_key(bound_value);
    while (hi - lo > 64u) {
        const uint32_t span = hi - lo;
        const uint32_t step = wpp_step(span);
        uint32_t c = 0;
        for (uint32_t lane = 0; lane < 64; lane++) {
            const uint64_t pidx = (uint64_t)lo + (uint64_t)(lane + 1) * step - 1;
            bool t = false;
            if (pidx < hi) {
                const int32_t k = order_key(a[pidx]);
                t = strict ? (k < bound) : (k <= bound);
            }
            c += t;
        }
        const uint64_t nhi = (uint64_t)lo + (uint64_t)(c + 1) * step - 1;
        const uint32_t new_lo = lo + c * step;
        hi = nhi < hi ? (uint32_t)nhi : hi;
        lo = new_lo;
    }
    uint32_t c = 0;
    for (uint32_t lane = 0; lane < 64; lane++)
        if (lo + lane < hi) {
            const int32_t k = order_key(a[lo + lane]);
            c += strict ? (k < bound) : (k <= bound);
        }
    return lo + c;
}

// select_peak_lut (the rescoring kernel's table-driven lookup) next to select_most_intense_peak on the same window
int emu_select_peak_lut(const float* masses, const float* intens, uint32_t n, float center, int kind, float tlo, float thi) {
    Tol t{kind, tlo, thi};
    const float w = peak_lut_width(n ? masses[n - 1] : 0.0f);
    std::vector<uint32_t> plut(PLUT_BINS);
    for (uint32_t b = 0; b < PLUT_BINS; b++) plut[b] = peak_lut_entry(masses, n, b, w);
    float lo, hi;
    tol_bounds(t, center, lo, hi);
    if (pow2_reciprocal(w) != 1.0f / w) return -2;  // (the kernel's reciprocal)
    return select_peak_lut(masses, intens, n, plut.data(), pow2_reciprocal(w), lo, hi);
}

// rescore_kernel's peak-presence filter (core.h: pbm_*) against the thing it must never contradict: for every ion and fragment
// charge 1..3, "some peak lies inside Tolerance::bounds(ion / charge)" implies "the bit of the ion's bin is set".  The bitmap is
// built as build_peak_bitmap (kernels.hip) builds it, the bins are taken the way the kernel takes them (one conversion of the
// ion, integer halves and thirds, modulo the bitmap).  Returns the number of violations; counts matches and set bits among the
// items for the statistics.
uint32_t emu_peak_bitmap_violations(const float* masses, uint32_t n, int kind, float tlo, float thi, const float* ions, uint32_t m,
                                    uint32_t* n_match, uint32_t* n_set, int* filter_active) {
    Tol t{kind, tlo, thi};
    const PbmReach reach = pbm_reach_of(t);  // (the scorer's, worked out on the host: DevScorer::pbm_reach)
    bool ok = true;
    for (uint32_t i = 0; i < n && ok; i++) {
        float D;
        ok = pbm_peak_reach(reach, masses[i], D);  // (a NaN or negative mass: the filter is switched off)
    }
    std::vector<uint32_t> bm(PBM_WORDS, ok ? 0u : 0xFFFFFFFFu);
    if (ok)
        for (uint32_t i = 0; i < n; i++) {
            float D;
            pbm_peak_reach(reach, masses[i], D);
            uint32_t b0, b1;
            pbm_peak_span(masses[i], D, b0, b1);
            for (uint32_t b = b0; b <= b1; b++) bm[(b & (PBM_BITS - 1u)) >> 5] |= 1u << (b & 31u);
        }
    *filter_active = ok ? 1 : 0;
    uint32_t bad = 0;
    *n_match = *n_set = 0;
    for (uint32_t j = 0; j < m; j++) {
        const uint32_t x = pbm_index(ions[j]);
        if (pbm_bin_c3(x) != ((x / 3u) & (PBM_BITS - 1u))) bad++;  // (the multiply-shift is a division by three)
        for (uint32_t c = 1; c <= 3; c++) {
            const float mz = c == 1 ? ions[j] : ions[j] / (float)c;  // scoring.rs:707: the reference's own quotient
            float lo, hi;
            tol_bounds(t, mz, lo, hi);
            bool match = false;
            for (uint32_t i = 0; i < n && !match; i++) match = masses[i] >= lo && masses[i] <= hi;  // spectrum.rs:147-157
            const uint32_t bin = c == 1 ? pbm_bin_c1(x) : c == 2 ? pbm_bin_c2(x) : pbm_bin_c3(x);
            const bool set = (bm[bin >> 5] >> (bin & 31u)) & 1u;
            *n_match += match;
            *n_set += set;
            bad += match && !set;
        }
    }
    return bad;
}

// The succinct form of a position-table row (core.h: LutWord / lut_rank; built the way index_build.hip builds it: occupancy bit of
// cell c = lut[c + 1] != lut[c], rank = exclusive count of the bits + `base`, pos = the run starts of the non-empty cells, then the
// tile's end) against the row itself: pos[rank(c)] must be lut[c] for EVERY cell c, the end marker included.  `mz`: one tile's
// entries, ascending in the total order.  Returns the number of cells where the two differ.
uint64_t emu_succinct_lut_mismatches(const float* mz, uint32_t n, float scale, uint32_t stride, uint32_t base, uint64_t* n_nonempty) {
    std::vector<uint32_t> lut(stride);
    for (uint32_t c = 0; c < stride; c++) lut[c] = lut_entry(mz, 1, 0, n, c, stride, scale);
    const uint32_t words = (stride + 31) / 32;
    std::vector<LutWord> l1(words);
    std::vector<uint32_t> pos;
    uint32_t r = base;
    for (uint32_t w = 0; w < words; w++) {
        uint32_t bits = 0;
        for (uint32_t b = 0; b < 32; b++) {
            const uint32_t c = w * 32 + b;
            if (c + 1 < stride && lut[c + 1] != lut[c]) bits |= 1u << b;
        }
        l1[w].bits = bits;
        l1[w].rank = r;
        for (uint32_t b = 0; b < 32; b++)
            if ((bits >> b) & 1u) { pos.push_back(lut[w * 32 + b]); r++; }
    }
    pos.push_back(lut[stride - 1]);
    *n_nonempty = pos.size() - 1;
    uint64_t bad = 0;
    for (uint32_t c = 0; c < stride; c++) bad += pos[lut_rank(l1[c >> 5], c) - base] != lut[c];
    return bad;
}

// Run::matched (scoring.rs:771-793) fed the same index sequence through core.h's four-field Run and the one-register form the
// rescoring kernel keeps: returns the number of steps after which `longest` differs (0 = equivalent on this sequence).
uint32_t emu_run_packed_mismatches(const uint32_t* indices, uint32_t n) {
    Run r{0, 0, 0, 0};
    uint32_t p = 0, bad = 0;
    for (uint32_t i = 0; i < n; i++) {
        run_matched(r, indices[i]);
        run_matched_packed(p, indices[i]);
        bad += run_longest_packed(p) != r.longest || (p & 1023u) != r.start + r.length || ((p >> 10) & 1023u) != r.length;
    }
    return bad;
}

// ... and the two-register form (21-bit fields) of the instance that scores databases with peptides of more than 1023 residues
uint32_t emu_run_packed64_mismatches(const uint32_t* indices, uint32_t n) {
    Run r{0, 0, 0, 0};
    uint64_t p = 0;
    uint32_t bad = 0;
    for (uint32_t i = 0; i < n; i++) {
        run_matched(r, indices[i]);
        run_matched_packed(p, indices[i]);
        bad += run_longest_packed(p) != r.longest || (uint32_t)(p & 0x1FFFFFu) != r.start + r.length || (uint32_t)((p >> 21) & 0x1FFFFFu) != r.length;
    }
    return bad;
}

}  // extern "C"
