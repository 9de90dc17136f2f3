// kernels.hip — gfx950 kernels of the search-and-score path.
//
//   prelim_kernel  : Scorer::initial_hits (scoring.rs:418-462) = precursor-window query
//                    (database.rs:402-425) + matched-fragment counting (scoring.rs:358-375 over
//                    database.rs:480-536) + the nested trim_hits k-selects (scoring.rs:322-329).
//   rescore_kernel : Scorer::build_features / score_candidate / score_chimera_fast
//                    (scoring.rs:478-595, 675-767, 648-672, 598-644).
//
// One 64-lane wavefront owns one spectrum.  Spectrum peaks and their fragment-tolerance windows live
// in LDS, candidate counters live in LDS (u16 pairs), the window's fragments are one contiguous
// range of the peptide-major index and are streamed with coalesced 8-byte loads.  This is sparse
// gather/compare/accumulate work: no MFMA.  Compile with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include "device_types.h"

using namespace sagecore;

namespace sagehip {

namespace {

constexpr uint32_t WAVE = 64;

// optional per-phase cycle accounting (DevWork::dbg != null): the first DBG_BLOCKS blocks of a launch
// store clock deltas into their own slot [block][kernel*8 + phase] (no atomics: nothing is perturbed)
constexpr uint32_t DBG_BLOCKS = 4096;
struct PhaseClock {
    unsigned long long* slot;
    long long t;
    __device__ __forceinline__ void start(unsigned long long* dbg, uint32_t blk, uint32_t kernel) {
        slot = (dbg && blk < DBG_BLOCKS) ? dbg + (size_t)blk * 16 + kernel * 8 : nullptr;
        if (slot) t = clock64();
    }
    __device__ __forceinline__ void mark(int phase) {
        if (slot) {
            const long long n = clock64();
            if ((threadIdx.x & 63u) == 0) slot[phase] += (unsigned long long)(n - t);
            t = n;
        }
    }
};

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// partition_point over sorted a[lo..hi) of key(a[i]) < bound (STRICT) or <= bound, all 64 lanes
// cooperating: 64 pivots per round (log_65 instead of log_2 dependent loads).
template <bool STRICT>
__device__ __forceinline__ uint32_t wave_partition_point(const float* __restrict__ a, uint32_t lo, uint32_t hi,
                                                         int32_t bound) {
    const uint32_t lane = lane_id();
    while (hi - lo > WAVE) {
        const uint32_t span = hi - lo;
        const uint32_t step = (span + WAVE) / (WAVE + 1);
        const uint64_t pidx = (uint64_t)lo + (uint64_t)(lane + 1) * step - 1;
        bool t = false;
        if (pidx < hi) {
            const int32_t k = order_key(a[pidx]);
            t = STRICT ? (k < bound) : (k <= bound);
        }
        const uint32_t c = (uint32_t)__popcll(__ballot(t));
        const uint64_t nhi = (uint64_t)lo + (uint64_t)(c + 1) * step - 1;
        const uint32_t new_lo = lo + c * step;
        hi = nhi < hi ? (uint32_t)nhi : hi;
        lo = new_lo;
    }
    bool t = false;
    if (lo + lane < hi) {
        const int32_t k = order_key(a[lo + lane]);
        t = STRICT ? (k < bound) : (k <= bound);
    }
    return lo + (uint32_t)__popcll(__ballot(t));
}

// ---- candidate counters ---------------------------------------------------------------------
// narrow path: u16 pairs in LDS;  large-window path: u32 in global scratch
template <bool WIDE>
struct Counters {
    uint32_t* p;
    __device__ __forceinline__ void zero(uint32_t n, uint32_t lane) {
        if (WIDE) {
            for (uint32_t i = lane; i < n; i += WAVE) p[i] = 0;
        } else {
            for (uint32_t i = lane; i < (n + 1) / 2; i += WAVE) p[i] = 0;
        }
    }
    __device__ __forceinline__ void add(uint32_t idx, uint32_t c) {
        if (WIDE) atomicAdd(&p[idx], c);
        else atomicAdd(&p[idx >> 1], c << ((idx & 1) * 16));
    }
    __device__ __forceinline__ uint32_t get(uint32_t idx) const {
        if (WIDE) return __hip_atomic_load(&p[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return (p[idx >> 1] >> ((idx & 1) * 16)) & 0xFFFFu;
    }
};

// ---- k-select state kept in registers: lane i holds heap element i (k <= 64) -------------------
// bounded_min_heapify (heap.rs:7-60) is inherently sequential, but every index it touches is
// wave-uniform, so the heap can live one element per lane and be driven with v_readlane/v_writelane
// (a few cycles each) instead of dependent LDS round trips.
struct WaveHeap {
    uint32_t lo, hi;
};
__device__ __forceinline__ uint64_t wh_get(const WaveHeap& h, uint32_t idx) {
    idx = __builtin_amdgcn_readfirstlane(idx);
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)h.hi, (int)idx) << 32) |
           (uint32_t)__builtin_amdgcn_readlane((int)h.lo, (int)idx);
}
__device__ __forceinline__ void wh_set(WaveHeap& h, uint32_t idx, uint64_t v) {
    const bool me = lane_id() == idx;  // v_writelane as compare + select (idx and v are wave-uniform)
    h.lo = me ? (uint32_t)v : h.lo;
    h.hi = me ? (uint32_t)(v >> 32) : h.hi;
}
__device__ __forceinline__ uint64_t lane_value(uint64_t v, uint32_t src_lane) {  // broadcast lane src_lane's v
    src_lane = __builtin_amdgcn_readfirstlane(src_lane);
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)src_lane) << 32) |
           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)src_lane);
}
__device__ __forceinline__ void wh_sift_down(WaveHeap& h, uint32_t len, uint32_t index) {  // heap.rs:40-60
    const uint64_t moving = wh_get(h, index);
    for (;;) {
        const uint32_t l = index * 2 + 1;
        if (l >= len) break;
        uint32_t smallest = index;
        uint64_t sv = moving;
        const uint64_t vl = wh_get(h, l);
        if (vl < sv) { smallest = l; sv = vl; }
        const uint32_t r = l + 1;
        if (r < len) {
            const uint64_t vr = wh_get(h, r);
            if (vr < sv) { smallest = r; sv = vr; }
        }
        if (smallest == index) break;
        wh_set(h, index, sv);       // slice.swap(smallest, index)
        wh_set(h, smallest, moving);
        index = smallest;
    }
}
__device__ __forceinline__ void wh_build(WaveHeap& h, uint32_t k) {  // heap.rs:13-15
    for (uint32_t i = k / 2; i-- > 0;) wh_sift_down(h, k, i);
}
__device__ __forceinline__ void wh_offer(WaveHeap& h, uint32_t k, uint64_t v) {  // heap.rs:21-27
    if (k && v > wh_get(h, 0)) {
        wh_set(h, 0, v);
        wh_sift_down(h, k, 0);
    }
}

// CList (core.h) with wave-uniform bookkeeping: every lane holds the same stored/len, appends are
// lane-parallel (ballot prefix), trims run on a WaveHeap.
struct UList {
    uint64_t* items;
    uint32_t stored, cap;
    uint64_t len;
    bool ok;
};
// append `nvalid` (<= 64) logical entries, lane i supplying entry i (clist_push's rule per entry)
__device__ __forceinline__ void ulist_append(UList& c, uint64_t v, uint32_t nvalid, uint32_t kmax) {
    const uint32_t lane = lane_id();
    const bool store = lane < nvalid && (c.len + lane < kmax || v != PRESCORE_EMPTY);
    const uint64_t sm = __ballot(store);
    const uint32_t pos = c.stored + (uint32_t)__popcll(sm & ((1ull << lane) - 1ull));
    if (store && pos < c.cap) c.items[pos] = v;
    const uint32_t total = c.stored + (uint32_t)__popcll(sm);
    c.ok = c.ok && total <= c.cap;
    c.stored = total <= c.cap ? total : c.cap;
    c.len += nvalid;
}
__device__ __forceinline__ void ulist_append_empties(UList& c, uint64_t n, uint32_t kmax) {  // clist_push_empties
    const uint64_t room = c.len < kmax ? kmax - c.len : 0;
    const uint32_t lit = (uint32_t)(n < room ? n : room);  // < kmax <= 64
    if (lit) ulist_append(c, PRESCORE_EMPTY, lit, kmax);
    c.len += n - lit;
}
// trim_hits (scoring.rs:322-329)
__device__ __forceinline__ void ulist_trim(UList& c, uint32_t report_psms) {
    const uint32_t lane = lane_id();
    const uint32_t k = trim_k(c.len, report_psms);
    if (c.len > k) {
        __syncthreads();
        WaveHeap h;
        const uint64_t mine = lane < k ? c.items[lane] : PRESCORE_EMPTY;
        h.lo = (uint32_t)mine;
        h.hi = (uint32_t)(mine >> 32);
        wh_build(h, k);
        for (uint32_t base = k; base < c.stored; base += WAVE) {
            const uint64_t v = base + lane < c.stored ? c.items[base + lane] : PRESCORE_EMPTY;
            const uint32_t n = c.stored - base < WAVE ? c.stored - base : WAVE;
            for (uint32_t j = 0; j < n; j++) wh_offer(h, k, lane_value(v, j));
        }
        __syncthreads();
        if (lane < k) c.items[lane] = ((uint64_t)h.hi << 32) | h.lo;
        __syncthreads();
    }
    c.stored = k;
    c.len = k;
}

struct PrelimLds {
    float* win_lo;
    float* win_hi;
    uint64_t* listA;
    uint64_t* listB;
    uint64_t* heap;
    uint32_t* cnt;
};

__device__ __forceinline__ PrelimLds carve_prelim(unsigned char* smem, const DevScorer& sc, const DevBatchView& b) {
    PrelimLds l;
    size_t off = 0;
    const bool fold = sc.min_isotope_err != sc.max_isotope_err;
    l.listA = (uint64_t*)(smem + off); off += fold ? (size_t)sc.list_cap * 8 : 0;
    l.listB = (uint64_t*)(smem + off); off += (size_t)sc.list_cap * 8;
    l.heap = (uint64_t*)(smem + off); off += (size_t)sc.kmax * 8;
    l.win_lo = (float*)(smem + off); off += (size_t)b.fzcap * b.pcap * 4;
    l.win_hi = (float*)(smem + off); off += (size_t)b.fzcap * b.pcap * 4;
    l.cnt = (uint32_t*)(smem + off);
    return l;
}

template <bool WIDE>
__global__ __launch_bounds__(64) void prelim_kernel(DevDbView db, DevScorer sc, DevBatchView b, DevWork w) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t lane = lane_id();
    const PrelimLds L = carve_prelim(smem, sc, b);
    Counters<WIDE> cnt;
    cnt.p = WIDE ? (w.wide_cnt + (size_t)blockIdx.x * w.wide_words) : L.cnt;
    if (WIDE && *w.n_deferred == 0) return;

    for (uint32_t blk = blockIdx.x; blk < b.n; blk += gridDim.x) {
        const uint32_t spec = b.order ? b.order[blk] : blk;
        if (WIDE && w.status[spec] != ST_DEFERRED) continue;
        __syncthreads();
        PhaseClock pc;
        pc.start(WIDE ? nullptr : w.dbg, blk, 0);
        const uint64_t p0 = b.peak_off[spec];
        const uint32_t P = (uint32_t)(b.peak_off[spec + 1] - p0);
        const float* __restrict__ masses = b.masses + p0;
        const uint32_t zraw = b.precursor_charge[spec];
        uint32_t z0, z1;
        if (sc.wide_window || zraw == 0 || sc.override_precursor_charge) {  // scoring.rs:423, 437, 442
            z0 = sc.min_precursor_charge;
            z1 = sc.max_precursor_charge;
        } else {
            z0 = z1 = zraw;
        }
        uint32_t nfz_max = 0;
        for (uint32_t z = z0; z <= z1; z++) {
            const uint32_t m = max_fragment_charge(sc.max_fragment_charge, z) - 1;
            nfz_max = m > nfz_max ? m : nfz_max;
        }
        if (nfz_max > b.fzcap) nfz_max = b.fzcap;  // (upload sized fzcap from the same rule)

        // fragment-tolerance window of every (peak, fragment charge): database.rs:481 on the
        // experimental mass peak*charge of scoring.rs:360
        for (uint32_t i = lane; i < P; i += WAVE) {
            const float m = masses[i];
            for (uint32_t fz = 1; fz <= nfz_max; fz++) {
                float lo, hi;
                tol_bounds(sc.fragment_tol, m * (float)fz, lo, hi);
                L.win_lo[(size_t)(fz - 1) * b.pcap + i] = lo;
                L.win_hi[(size_t)(fz - 1) * b.pcap + i] = hi;
            }
        }
        __syncthreads();
        bool mono_ok = true;  // the sorted-count shortcut needs ascending bounds and lo <= hi
        for (uint32_t fz = 0; fz < nfz_max; fz++) {
            const float* wl = L.win_lo + (size_t)fz * b.pcap;
            const float* wh = L.win_hi + (size_t)fz * b.pcap;
            for (uint32_t i = lane; i < P; i += WAVE) {
                mono_ok = mono_ok && (wl[i] <= wh[i]);
                if (i > 0) mono_ok = mono_ok && (wl[i - 1] <= wl[i]) && (wh[i - 1] <= wh[i]);
            }
        }
        const bool sorted_ok = __ballot(!mono_ok) == 0ull;
        const uint32_t ptop = pow2_floor(P);
        pc.mark(0);

        const float mzp = b.precursor_mz[spec] - PROTON;  // scoring.rs:420
        Tol iso_tol;
        iso_tol.kind = 2;
        iso_tol.lo = -2.4f;
        iso_tol.hi = 2.4f;  // scoring.rs:430
        if (b.isolation_lo && b.isolation_hi) {
            const float a = b.isolation_lo[spec], c = b.isolation_hi[spec];
            if (a == a && c == c) { iso_tol.lo = a; iso_tol.hi = c; }
        }
        const bool fold = sc.min_isotope_err != sc.max_isotope_err;  // scoring.rs:391
        const int isoA = fold ? sc.min_isotope_err : 0, isoB = fold ? sc.max_isotope_err : 0;

        UList A, B;  // wave-uniform state
        A.items = L.listA; A.cap = fold ? sc.list_cap : 0; A.stored = 0; A.len = 0; A.ok = true;
        B.items = L.listB; B.cap = sc.list_cap; B.stored = 0; B.len = 0; B.ok = true;
        bool deferred = false, deferred_open = false;  // uniform
        uint32_t tot_matched = 0, tot_scored = 0;  // uniform

        for (uint32_t z = z0; z <= z1 && !deferred; z++) {
            const uint32_t nfz = max_fragment_charge(sc.max_fragment_charge, z) - 1;
            const float precursor_mass = mzp * (float)z;
            const Tol ptol = sc.wide_window ? tol_scaled(iso_tol, (float)z) : sc.precursor_tol;
            if (fold) { A.stored = 0; A.len = 0; }
            for (int iso = isoA; iso <= isoB && !deferred; iso++) {
                // ---- IndexedDatabase::query, database.rs:402-425 ----
                const float center = precursor_mass - (float)iso * NEUTRON;  // scoring.rs:344
                float plo, phi;
                tol_bounds(ptol, center, plo, phi);
                uint32_t left = wave_partition_point<true>(db.pep_mono, 0, db.np, order_key(plo));
                left = left ? left - 1 : 0;
                // the window is short in a narrow search: bracket it by galloping from `left` before searching
                uint32_t ghi = left;
                for (uint64_t span = WAVE;; span *= 16) {
                    ghi = (uint64_t)left + span < db.np ? (uint32_t)(left + span) : db.np;
                    if (ghi == db.np || order_key(db.pep_mono[ghi - 1]) > order_key(phi)) break;
                }
                const uint32_t right = wave_partition_point<false>(db.pep_mono, left, ghi, order_key(phi));
                const uint32_t potential = right - left + 1;  // scoring.rs:351
                if (!WIDE && potential > sc.wcap) {
                    deferred = true;
                    deferred_open = potential > sc.open_thresh && w.open_blocks > 0;
                    break;
                }
                if (WIDE && potential > w.wide_words) {  // a later query of this spectrum is an open-search window
                    deferred = true;
                    deferred_open = true;
                    break;
                }
                pc.mark(1);
                cnt.zero(potential, lane);
                // edge rule of database.rs:526-531: interior indices are in range by construction
                uint32_t first = left, end = right;
                if (left < db.np && !(db.pep_mono[left] >= plo)) first = left + 1;
                if (right < db.np && db.pep_mono[right] <= phi) end = right + 1;
                __syncthreads();
                uint32_t acc = 0;
                if (first < end) {
                    const uint64_t f0 = db.pm_off[first], f1 = db.pm_off[end];
                    auto count_one = [&](float frag) -> uint32_t {
                        if (!sorted_ok) {
                            uint32_t c = 0;
                            for (uint32_t fz = 0; fz < nfz; fz++)
                                c += count_windows_scan(L.win_lo + (size_t)fz * b.pcap, L.win_hi + (size_t)fz * b.pcap, P, frag);
                            return c;
                        }
                        switch (nfz) {
                            case 1: return count_windows_lockstep<1>(L.win_lo, L.win_hi, b.pcap, P, ptop, frag);
                            case 2: return count_windows_lockstep<2>(L.win_lo, L.win_hi, b.pcap, P, ptop, frag);
                            case 3: return count_windows_lockstep<3>(L.win_lo, L.win_hi, b.pcap, P, ptop, frag);
                            default: {
                                uint32_t c = 0;
                                for (uint32_t fz = 0; fz < nfz; fz++)
                                    c += count_windows_sorted(L.win_lo + (size_t)fz * b.pcap, L.win_hi + (size_t)fz * b.pcap, P, frag);
                                return c;
                            }
                        }
                    };
                    // software-pipelined stream over the window's fragments: the next trip's two 8-byte
                    // loads are issued before this trip's LDS searches
                    SageTheoretical n0{0, 0.f}, n1{0, 0.f};
                    uint64_t j = f0 + lane;
                    if (j < f1) n0 = db.pm_frag[j];
                    if (j + WAVE < f1) n1 = db.pm_frag[j + WAVE];
                    for (; j < f1; j += 2 * WAVE) {
                        const SageTheoretical fr0 = n0, fr1 = n1;
                        const bool has1 = j + WAVE < f1;
                        const uint64_t jn = j + 2 * WAVE;
                        if (jn < f1) n0 = db.pm_frag[jn];
                        if (jn + WAVE < f1) n1 = db.pm_frag[jn + WAVE];
                        const uint32_t c0 = count_one(fr0.fragment_mz);
                        const uint32_t c1 = has1 ? count_one(fr1.fragment_mz) : 0;
                        if (c0) { cnt.add(fr0.peptide_index - left, c0); acc += c0; }
                        if (c1) { cnt.add(fr1.peptide_index - left, c1); acc += c1; }
                    }
                }
                const uint32_t matched = wave_sum(acc);
                __syncthreads();
                pc.mark(2);
                tot_matched += matched;
                UList& target = fold ? A : B;
                if (matched == 0) {  // scoring.rs:376-378: the untrimmed all-default vector
                    ulist_append_empties(target, potential, sc.kmax);
                    continue;
                }
                // ---- trim_hits of this query, scoring.rs:380 ----
                const uint32_t k = trim_k(potential, sc.report_psms);
                uint32_t scored = 0;
                if (potential <= k) {  // no k-select: the slots go to the list verbatim
                    for (uint32_t base = 0; base < potential; base += WAVE) {
                        const uint32_t i = base + lane;
                        const uint32_t c = i < potential ? cnt.get(i) : 0;
                        scored += (uint32_t)__popcll(__ballot(c > 0));
                        const uint32_t nvalid = potential - base < WAVE ? potential - base : WAVE;
                        ulist_append(target, c ? pack_prescore(c, left + i, z, iso) : PRESCORE_EMPTY, nvalid, sc.kmax);
                    }
                } else {
                    WaveHeap h;
                    {
                        const uint32_t c = lane < k ? cnt.get(lane) : 0;
                        const uint64_t v = c ? pack_prescore(c, left + lane, z, iso) : PRESCORE_EMPTY;
                        h.lo = (uint32_t)v;
                        h.hi = (uint32_t)(v >> 32);
                        scored += (uint32_t)__popcll(__ballot(c > 0));
                    }
                    wh_build(h, k);
                    for (uint32_t base = k; base < potential; base += WAVE) {
                        const uint32_t i = base + lane;
                        const uint32_t c = i < potential ? cnt.get(i) : 0;
                        const uint64_t v = pack_prescore(c, left + i, z, iso);
                        uint64_t mask = __ballot(c > 0);
                        scored += (uint32_t)__popcll(mask);
                        while (mask) {  // in slot order; empty slots can never displace the heap minimum
                            const uint32_t bit = (uint32_t)__ffsll((long long)mask) - 1;
                            mask &= mask - 1;
                            wh_offer(h, k, lane_value(v, bit));
                        }
                    }
                    ulist_append(target, ((uint64_t)h.hi << 32) | h.lo, k, sc.kmax);
                }
                tot_scored += scored;
                __syncthreads();
                pc.mark(3);
            }
            if (fold && !deferred) {  // scoring.rs:405 then `hits +=` at :432 / :450
                ulist_trim(A, sc.report_psms);
                __syncthreads();
                for (uint32_t base = 0; base < A.stored; base += WAVE) {
                    const uint64_t v = base + lane < A.stored ? A.items[base + lane] : PRESCORE_EMPTY;
                    const uint32_t nvalid = A.stored - base < WAVE ? A.stored - base : WAVE;
                    ulist_append(B, v, nvalid, sc.kmax);
                }
                __syncthreads();
            }
        }
        if (deferred) {
            if (lane == 0) {
                w.status[spec] = deferred_open ? ST_DEFERRED_OPEN : ST_DEFERRED;
                atomicAdd(w.n_deferred + (deferred_open ? 2 : 0), 1u);
            }
            continue;
        }
        ulist_trim(B, sc.report_psms);  // scoring.rs:460
        __syncthreads();
        if (lane == 0) {
            if (!(A.ok && B.ok)) atomicAdd(w.n_deferred + 1, 1u);
            w.status[spec] = (A.ok && B.ok) ? ST_OK : ST_OVERFLOW;
            w.cand_len[spec] = B.stored;
            w.totals[2 * spec] = tot_matched;
            w.totals[2 * spec + 1] = tot_scored;
        }
        for (uint32_t i = lane; i < B.stored; i += WAVE) w.cand[(size_t)spec * sc.kmax + i] = L.listB[i];
        pc.mark(4);
    }
}


// ---- open / wide-window searches ------------------------------------------------------------------
// When the precursor window holds 10^4..10^6 candidates, streaming all of their fragments (peptide-major)
// is the wrong loop order.  Here every (peak, fragment charge) window is looked up in the m/z-major copy
// of the index: one table lookup gives a start position, then the wavefront streams the few hundred
// fragments of that m/z window with coalesced 8-byte loads and bumps a u16 counter per in-window
// peptide.  Counters live in a per-block slab in HBM that is all-zero between spectra: the k-select
// scan clears what it reads.  Same predicate as database.rs:526-533, so the counts are identical.
__global__ __launch_bounds__(64) void prelim_open_kernel(DevDbView db, DevScorer sc, DevBatchView b, DevWork w) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t lane = lane_id();
    if (w.n_deferred[2] == 0) return;
    const bool fold = sc.min_isotope_err != sc.max_isotope_err;  // scoring.rs:391
    uint64_t* listA = (uint64_t*)smem;
    uint64_t* listB = listA + (fold ? sc.list_cap : 0);
    uint32_t* cnt = w.open_cnt + (size_t)blockIdx.x * w.open_words;

    for (uint32_t blk = blockIdx.x; blk < b.n; blk += gridDim.x) {
        const uint32_t spec = b.order ? b.order[blk] : blk;
        if (w.status[spec] != ST_DEFERRED_OPEN) continue;
        __syncthreads();
        PhaseClock pc;
        pc.start(w.dbg, blk, 0);
        const uint64_t p0 = b.peak_off[spec];
        const uint32_t P = (uint32_t)(b.peak_off[spec + 1] - p0);
        const float* __restrict__ masses = b.masses + p0;
        const uint32_t zraw = b.precursor_charge[spec];
        uint32_t z0, z1;
        if (sc.wide_window || zraw == 0 || sc.override_precursor_charge) {
            z0 = sc.min_precursor_charge;
            z1 = sc.max_precursor_charge;
        } else {
            z0 = z1 = zraw;
        }
        const float mzp = b.precursor_mz[spec] - PROTON;  // scoring.rs:420
        Tol iso_tol;
        iso_tol.kind = 2;
        iso_tol.lo = -2.4f;
        iso_tol.hi = 2.4f;  // scoring.rs:430
        if (b.isolation_lo && b.isolation_hi) {
            const float a = b.isolation_lo[spec], c = b.isolation_hi[spec];
            if (a == a && c == c) { iso_tol.lo = a; iso_tol.hi = c; }
        }
        const int isoA = fold ? sc.min_isotope_err : 0, isoB = fold ? sc.max_isotope_err : 0;
        UList A, B;
        A.items = listA; A.cap = fold ? sc.list_cap : 0; A.stored = 0; A.len = 0; A.ok = true;
        B.items = listB; B.cap = sc.list_cap; B.stored = 0; B.len = 0; B.ok = true;
        uint32_t tot_matched = 0, tot_scored = 0;

        for (uint32_t z = z0; z <= z1; z++) {
            const uint32_t nfz = max_fragment_charge(sc.max_fragment_charge, z) - 1;
            const float precursor_mass = mzp * (float)z;
            const Tol ptol = sc.wide_window ? tol_scaled(iso_tol, (float)z) : sc.precursor_tol;
            if (fold) { A.stored = 0; A.len = 0; }
            for (int iso = isoA; iso <= isoB; iso++) {
                const float center = precursor_mass - (float)iso * NEUTRON;  // scoring.rs:344
                float plo, phi;
                tol_bounds(ptol, center, plo, phi);
                uint32_t left = wave_partition_point<true>(db.pep_mono, 0, db.np, order_key(plo));
                left = left ? left - 1 : 0;
                const uint32_t right = wave_partition_point<false>(db.pep_mono, left, db.np, order_key(phi));
                const uint32_t potential = right - left + 1;  // scoring.rs:351
                uint32_t first = left, end = right;           // database.rs:526-531
                if (left < db.np && !(db.pep_mono[left] >= plo)) first = left + 1;
                if (right < db.np && db.pep_mono[right] <= phi) end = right + 1;

                // ---- matched-fragment counting, scoring.rs:358-375 ----
                pc.mark(5);
                uint32_t acc = 0;
                if (first < end) {
                    for (uint32_t i = 0; i < P; i++) {
                        const float m = masses[i];
                        for (uint32_t fz = 1; fz <= nfz; fz++) {
                            float lo, hi;
                            tol_bounds(sc.fragment_tol, m * (float)fz, lo, hi);
                            // conservative start: one table cell below the cell of `lo`
                            float cell = floorf(lo * db.lut_scale) - 1.0f;
                            cell = cell > 0.0f ? cell : 0.0f;  // also maps NaN to 0
                            const uint32_t bin = cell < (float)(db.lut_n - 1) ? (uint32_t)cell : db.lut_n - 1;
                            // four 8-byte loads per lane are in flight per trip (2 KiB per wavefront)
                            for (uint64_t j = (uint64_t)db.mz_lut[bin] + lane;; j += 4 * WAVE) {
                                SageTheoretical fr[4];
                                bool in[4];
#pragma unroll
                                for (int q = 0; q < 4; q++) {
                                    in[q] = j + (uint64_t)q * WAVE < db.nf;
                                    fr[q] = SageTheoretical{0, 0.f};
                                    if (in[q]) fr[q] = db.mz_frag[j + (uint64_t)q * WAVE];
                                }
#pragma unroll
                                for (int q = 0; q < 4; q++) {
                                    const bool hit = in[q] && fr[q].fragment_mz >= lo && fr[q].fragment_mz <= hi &&
                                                     fr[q].peptide_index >= first && fr[q].peptide_index < end;
                                    if (hit) {
                                        const uint32_t idx = fr[q].peptide_index - left;
                                        if (!(sc.dbg_flags & 2)) atomicAdd(&cnt[idx >> 1], 1u << ((idx & 1) * 16));
                                        acc++;
                                    }
                                }
                                // ascending m/z: done once the last quarter holds nothing at or below `hi`
                                if (__ballot(in[3] && !(fr[3].fragment_mz > hi)) == 0ull) break;
                            }
                        }
                    }
                }
                const uint32_t matched = wave_sum(acc);
                // the counters are only ever touched by this wavefront, through L2 (atomics, sc1 loads, write-through
                // stores): ordering needs the earlier operations to have completed, not an L2 write-back
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                pc.mark(6);
                tot_matched += matched;
                UList& target = fold ? A : B;
                if (matched == 0) {  // scoring.rs:376-378 (no counter was touched: the slab is still zero)
                    ulist_append_empties(target, potential, sc.kmax);
                    continue;
                }
                // ---- trim_hits, scoring.rs:380: one pass over the slots, 4 per lane, clearing as it goes ----
                const uint32_t k = trim_k(potential, sc.report_psms);
                const bool select = potential > k;
                WaveHeap h{0, 0};
                uint32_t scored = 0;
                for (uint32_t base = 0; base < ((sc.dbg_flags & 4) ? 256u : potential); base += 4 * WAVE) {
                    const uint32_t s0 = base + 4 * lane;  // this lane's first slot
                    uint32_t* wp = cnt + (s0 >> 1);
                    const uint32_t w0 = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t w1 = __hip_atomic_load(wp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (w0) wp[0] = 0;
                    if (w1) wp[1] = 0;
                    uint32_t c[4] = {w0 & 0xFFFFu, w0 >> 16, w1 & 0xFFFFu, w1 >> 16};
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        if (s0 + q >= potential) c[q] = 0;
                        scored += (uint32_t)__popcll(__ballot(c[q] > 0));
                    }
                    if (!select) {  // potential <= k <= 64: every slot goes to the list verbatim, in slot order
                        // slot i sits in lane i/4, sub i%4
                        uint32_t ci = 0;
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const uint32_t v = __shfl(c[q], (int)(lane >> 2), 64);
                            if ((lane & 3u) == (uint32_t)q) ci = v;
                        }
                        const uint32_t nvalid = potential - base < WAVE ? potential - base : WAVE;
                        ulist_append(target, ci ? pack_prescore(ci, left + base + lane, z, iso) : PRESCORE_EMPTY, nvalid, sc.kmax);
                        continue;  // (potential <= 64 => single trip)
                    }
                    uint32_t from = 0;  // first slot of this chunk that is offered (slots < k seed the heap)
                    if (base == 0) {
                        uint32_t ci = 0;
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const uint32_t v = __shfl(c[q], (int)(lane >> 2), 64);
                            if ((lane & 3u) == (uint32_t)q) ci = v;
                        }
                        const uint64_t v = (lane < k && ci) ? pack_prescore(ci, left + lane, z, iso) : PRESCORE_EMPTY;
                        h.lo = (uint32_t)v;
                        h.hi = (uint32_t)(v >> 32);
                        wh_build(h, k);
                        from = k;
                    }
                    // offers in slot order; a slot can only enter if its count reaches the heap minimum's
                    const uint32_t hmin = prescore_matched(wh_get(h, 0));
                    bool cand = false;
#pragma unroll
                    for (int q = 0; q < 4; q++) cand = cand || (c[q] > 0 && c[q] >= hmin && s0 + q >= from);
                    uint64_t mask = __ballot(cand);
                    if (sc.dbg_flags & 1) mask = 0;
                    while (mask) {
                        const uint32_t src = (uint32_t)__ffsll((long long)mask) - 1;
                        mask &= mask - 1;
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const uint32_t cq = (uint32_t)__builtin_amdgcn_readlane((int)c[q], (int)__builtin_amdgcn_readfirstlane(src));
                            const uint32_t slot = base + 4 * src + q;
                            if (cq && slot >= from) wh_offer(h, k, pack_prescore(cq, left + slot, z, iso));
                        }
                    }
                }
                if (select) ulist_append(target, ((uint64_t)h.hi << 32) | h.lo, k, sc.kmax);
                tot_scored += scored;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                pc.mark(7);
            }
            if (fold) {  // scoring.rs:405 then `hits +=`
                ulist_trim(A, sc.report_psms);
                __syncthreads();
                for (uint32_t base = 0; base < A.stored; base += WAVE) {
                    const uint64_t v = base + lane < A.stored ? A.items[base + lane] : PRESCORE_EMPTY;
                    const uint32_t nvalid = A.stored - base < WAVE ? A.stored - base : WAVE;
                    ulist_append(B, v, nvalid, sc.kmax);
                }
                __syncthreads();
            }
        }
        ulist_trim(B, sc.report_psms);  // scoring.rs:460
        __syncthreads();
        if (lane == 0) {
            if (!(A.ok && B.ok)) atomicAdd(w.n_deferred + 1, 1u);
            w.status[spec] = (A.ok && B.ok) ? ST_OK : ST_OVERFLOW;
            w.cand_len[spec] = B.stored;
            w.totals[2 * spec] = tot_matched;
            w.totals[2 * spec + 1] = tot_scored;
        }
        for (uint32_t i = lane; i < B.stored; i += WAVE) w.cand[(size_t)spec * sc.kmax + i] = listB[i];
    }
}

// ---- rescoring -------------------------------------------------------------------------------
__device__ __forceinline__ double lnfact_dev(uint32_t n, const double* __restrict__ table, uint32_t table_n) {
    if (n < table_n) return table[n];
    const double x = (double)n;  // scoring.rs:170-177
    return x * log(x) - x + 0.5 * log(x) + 0.5 * log(3.14159265358979323846 * 2.0 * x);
}

__device__ __forceinline__ double hyperscore_dev(int score_type, const Score& s, const double* table, uint32_t tn) {
    double score;  // ScoreType::score, scoring.rs:179-201
    if (score_type == 0) {
        const double i = (double)(s.summed_b + 1.0f) * (double)(s.summed_y + 1.0f);
        score = log(i) + lnfact_dev(s.matched_b, table, tn) + lnfact_dev(s.matched_y, table, tn);
    } else {
        const float si = s.summed_b + s.summed_y;
        score = (double)log1pf(si) + lnfact_dev(s.matched_b, table, tn) + lnfact_dev(s.matched_y, table, tn);
    }
    return __builtin_isfinite(score) ? score : 255.0;
}

// Rescoring is split in two phases per candidate chunk so that all 64 lanes stay busy and the
// order-sensitive f32 sums still run in the reference's (kind, index, charge) order:
//   A. every (candidate, ion, fragment charge) item of the chunk is matched in parallel
//      (select_most_intense_peak, spectrum.rs:134-159) -> res[item] = peak index or NONE, in LDS;
//   B. one lane per candidate walks ITS items in order and accumulates (scoring.rs:704-754).
constexpr uint16_t RES_NONE = 0xFFFFu;

__global__ __launch_bounds__(64) void rescore_kernel(DevDbView db, DevScorer sc, DevBatchView b, DevWork w,
                                                     const double* __restrict__ lnfact_table, uint32_t lnfact_n,
                                                     uint32_t tcap, SageFeature* __restrict__ out,
                                                     uint32_t* __restrict__ out_count) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t lane = lane_id();
    if (blockIdx.x >= b.n) return;
    const uint32_t spec = b.order ? b.order[blockIdx.x] : blockIdx.x;
    // LDS carve
    double* s_sorted = (double*)smem;                         // [64] hyperscores by rank
    long long* s_key = (long long*)(smem + 64 * 8);           // [64] sort keys by lane
    unsigned long long* s_ionbase = (unsigned long long*)(smem + 128 * 8);  // [64] ion table offset per candidate
    uint32_t* s_incl = (uint32_t*)(smem + 192 * 8);           // [64] inclusive item prefix per candidate
    uint32_t* s_nfz = s_incl + 64;                            // [64] fragment charges per candidate
    float* pm = (float*)(s_nfz + 64);                         // [pcap] peak masses
    float* pi = pm + b.pcap;                                  // [pcap] peak intensities
    float* term = pi + b.pcap;                                // [tcap] ppm term per matched item
    uint16_t* res = (uint16_t*)(term + tcap);                 // [tcap] matched peak index per item
    uint8_t* rm = (uint8_t*)(res + tcap);                     // [pcap] chimera: peak selected by the winner
    uint8_t* rm2 = rm + b.pcap;

    if (w.status[spec] != ST_OK) {
        if (lane == 0) out_count[spec] = 0;
        return;
    }
    PhaseClock pc;
    pc.start(w.dbg, blockIdx.x, 1);
    const uint64_t p0 = b.peak_off[spec];
    uint32_t P = (uint32_t)(b.peak_off[spec + 1] - p0);
    for (uint32_t i = lane; i < P; i += WAVE) {
        pm[i] = b.masses[p0 + i];
        pi[i] = b.intensities[p0 + i];
    }
    float tic = b.tic[spec];
    const uint32_t ncand = w.cand_len[spec];
    const uint64_t mine = lane < ncand ? w.cand[(size_t)spec * sc.kmax + lane] : PRESCORE_EMPTY;
    const uint32_t pep = prescore_peptide(mine);
    const bool valid = pep != 0xFFFFFFFFu;  // scoring.rs:489
    const uint32_t z = prescore_charge(mine);
    const int iso = prescore_iso(mine);
    const uint32_t mfc = max_fragment_charge(sc.max_fragment_charge, z);
    const uint32_t nfz = mfc - 1;
    uint64_t ion_base = 0;
    uint32_t lm1 = 0, info = 0;
    float calc = 0.f;
    if (valid) {
        const uint64_t o1 = db.ion_off[pep + 1];
        ion_base = db.ion_off[pep];
        lm1 = db.n_kinds ? (uint32_t)((o1 - ion_base) / db.n_kinds) : 0;
        info = db.pep_info[pep];
        calc = db.pep_mono[pep];
    }
    // item bookkeeping: n_items per candidate and its inclusive prefix over lanes
    const uint32_t n_items = valid ? db.n_kinds * lm1 * nfz : 0;
    uint32_t incl = n_items;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off, 64);
        if ((int)lane >= off) incl += t;
    }
    const uint32_t excl = incl - n_items;
    s_incl[lane] = incl;
    s_ionbase[lane] = ion_base;
    s_nfz[lane] = nfz ? nfz : 1;

    const double lambda = (double)w.totals[2 * spec] / (double)w.totals[2 * spec + 1];  // scoring.rs:499
    const float mzp = b.precursor_mz[spec] - PROTON;                                   // scoring.rs:502
    const float rt = b.rt ? b.rt[spec] : 0.0f;
    float ims = 0.0f;
    if (b.ims) { const float v = b.ims[spec]; ims = v == v ? v : 0.0f; }
    const uint32_t fid = b.file_id ? b.file_id[spec] : 0;
    const uint32_t total_items = __shfl(incl, 63, 64);
    __syncthreads();
    pc.mark(0);

    const uint32_t rounds = sc.chimera ? sc.report_psms : 1;
    const uint32_t per_round = sc.chimera ? 1 : sc.report_psms;
    uint32_t n_emitted = 0;
    for (uint32_t round = 0; round < rounds; round++) {
        const uint32_t ptop = pow2_floor(P);
        Score s;
        s.peptide = pep;
        s.precursor_charge = z;
        s.isotope_error = iso;
        s.matched_b = s.matched_y = 0;
        s.summed_b = s.summed_y = 0.0f;
        s.ppm_difference = 0.0f;
        s.longest_b = s.longest_y = 0;
        // ---- score_candidate over chunks of candidates whose items fit in res[] ----
        uint32_t base = 0;  // item offset where the current chunk starts
        while (base < total_items) {
            // chunk = candidates with excl >= base and incl <= base + tcap (contiguous lanes)
            const bool in_chunk = n_items && excl >= base && incl - base <= tcap;
            uint32_t chunk_items = in_chunk ? incl - base : 0;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const uint32_t o = __shfl_xor(chunk_items, off, 64);
                chunk_items = o > chunk_items ? o : chunk_items;
            }
            if (chunk_items == 0) break;  // a single candidate larger than tcap cannot happen (tcap >= max items)
            // phase A: the ion mass of the NEXT item is fetched before the current one is matched, so the
            // global-load latency overlaps the LDS searches
            auto locate = [&](uint32_t t, uint32_t& charge) -> float {
                const uint32_t g = base + t;  // global item index
                uint32_t c = 0;               // first candidate with incl > g
#pragma unroll
                for (uint32_t step = 32; step; step >>= 1) {
                    const uint32_t probe = c + step;
                    c = (probe <= 64 && s_incl[probe - 1] <= g) ? probe : c;
                }
                const uint32_t local = g - (c ? s_incl[c - 1] : 0);
                const uint32_t cz = s_nfz[c];
                const uint32_t ion = local / cz;
                charge = local - ion * cz + 1;
                return db.ions[s_ionbase[c] + ion];
            };
            // A0: gather every item's ion mass into term[] first — the loads of different trips are
            //     independent, so several are in flight per lane; A1 then matches out of LDS and overwrites
            //     term[t] (read and written by the same lane) with the ppm term.
#pragma unroll 4
            for (uint32_t t = lane; t < chunk_items; t += WAVE) {
                uint32_t charge;
                term[t] = locate(t, charge);
                res[t] = (uint16_t)charge;
            }
            __syncthreads();
            for (uint32_t t = lane; t < chunk_items; t += WAVE) {
                const float mz = term[t] / (float)res[t];
                const int pk = select_most_intense_peak_lockstep(pm, pi, P, ptop, mz, sc.fragment_tol);
                if (pk >= 0) {
                    const float peak_mass = pm[pk];
                    res[t] = (uint16_t)pk;
                    // the per-match ppm term of scoring.rs:719-720; only its accumulation is order-sensitive
                    term[t] = pi[pk] * __builtin_fabsf(mz - peak_mass) * 2E6f / (mz + peak_mass);
                } else {
                    res[t] = RES_NONE;
                }
            }
            __syncthreads();
            pc.mark(1);
            // phase B
            if (in_chunk) {
                Run b_run = {0, 0, 0, 0}, y_run = {0, 0, 0, 0};
                uint32_t t = excl - base;
                for (uint32_t k = 0; k < db.n_kinds; k++) {
                    const bool nterm_kind = db.ion_kinds[k] <= 2;
                    for (uint32_t idx = 0; idx < lm1; idx++) {
                        for (uint32_t c = 1; c < mfc; c++, t++) {
                            const uint16_t r = res[t];
                            if (r == RES_NONE) continue;
                            const float peak_intensity = pi[r];
                            s.ppm_difference += term[t];
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
            }
            __syncthreads();
            pc.mark(2);
            base += chunk_items;
        }
        double h = 0.0;
        bool pass = false;
        if (valid) {
            s.ppm_difference /= s.summed_b + s.summed_y;  // scoring.rs:759
            h = hyperscore_dev(sc.score_type, s, lnfact_table, lnfact_n);
            pass = (s.matched_b + s.matched_y) >= sc.min_matched_peaks;  // scoring.rs:491
        }
        // stable sort, descending by hyperscore.total_cmp (scoring.rs:495), as a rank computation
        const long long key = order_key64(h);
        s_key[lane] = key;
        const uint64_t pmask = __ballot(pass);
        const uint32_t npass = (uint32_t)__popcll(pmask);
        __syncthreads();
        uint32_t rank = 0;
        if (pass) {
            uint64_t m = pmask;
            while (m) {
                const uint32_t j = (uint32_t)__ffsll((long long)m) - 1;
                m &= m - 1;
                const long long kj = s_key[j];
                rank += (kj > key) || (kj == key && j < lane);
            }
            s_sorted[rank] = h;
        }
        __syncthreads();
        pc.mark(3);
        if (pass && rank < per_round) {  // scoring.rs:504-594
            const double next = rank + 1 < npass ? s_sorted[rank + 1] : 0.0;
            const double best = s_sorted[0];
            const float precursor_mass = mzp * (float)z;
            const uint32_t k = s.matched_b + s.matched_y;
            const double log10_poisson =
                ((double)k * log(lambda) - lambda - lnfact_dev(k, lnfact_table, lnfact_n)) / 2.302585092994046;
            const float isotope_error = (float)iso * NEUTRON;
            const float delta_mass =
                (precursor_mass - calc - isotope_error) * 2E6f / (precursor_mass - isotope_error + calc);
            const uint32_t plen = info & 0xFFFF;
            SageFeature f;
            f.spec_index = spec;
            f.peptide_idx = pep;
            f.rank = sc.chimera ? round + 1 : rank + 1;  // scoring.rs:541, 664
            f.label = ((info >> 16) & 0xFF) ? -1 : 1;
            f.expmass = precursor_mass;
            f.calcmass = calc;
            f.rt = rt;
            f.ims = ims;
            f.delta_mass = delta_mass;
            f.isotope_error = isotope_error;
            f.average_ppm = s.ppm_difference;
            f.longest_y_pct = (float)s.longest_y / (float)plen;
            f.matched_intensity_pct = 100.0f * (s.summed_b + s.summed_y) / tic;
            f.ms2_intensity = s.summed_b + s.summed_y;
            f.hyperscore = h;
            f.delta_next = h - next;
            f.delta_best = best - h;
            f.poisson = __builtin_isfinite(log10_poisson) ? log10_poisson : -__builtin_huge_val();
            f.matched_peaks = k;
            f.longest_b = s.longest_b;
            f.longest_y = s.longest_y;
            f.scored_candidates = w.totals[2 * spec + 1];
            f.peptide_len = plen;
            f.file_id = fid;
            f.charge = (uint8_t)z;
            f.missed_cleavages = (uint8_t)(info >> 24);
            for (int q = 0; q < 6; q++) f.pad[q] = 0;
            out[(size_t)spec * sc.report_psms + (sc.chimera ? round : rank)] = f;
        }
        pc.mark(4);
        const uint32_t emitted = npass < per_round ? npass : per_round;
        n_emitted += emitted;
        if (!sc.chimera || emitted == 0 || round + 1 == rounds) break;

        // ---- remove_matched_peaks(winner), scoring.rs:598-644 ----
        const uint64_t wmask = __ballot(pass && rank == 0);
        const uint32_t wl = (uint32_t)__ffsll((long long)wmask) - 1;
        const uint32_t wmfc = __shfl(mfc, wl, 64);
        const uint32_t w_items = __shfl(n_items, wl, 64);
        const unsigned long long w_base = s_ionbase[wl];
        const float* wions = db.ions + w_base;
        for (uint32_t i = lane; i < P; i += WAVE) rm[i] = 0;
        __syncthreads();
        for (uint32_t t = lane; t < w_items; t += WAVE) {
            const uint32_t ion = t / (wmfc - 1), charge = t % (wmfc - 1) + 1;
            const int pk = select_most_intense_peak_lockstep(pm, pi, P, ptop, wions[ion] / (float)charge, sc.fragment_tol);
            if (pk >= 0) rm[pk] = 1;
        }
        __syncthreads();
        // `to_remove.contains(&(mass, intensity))` compares values: equal pairs go together
        for (uint32_t i = lane; i < P; i += WAVE) {
            uint8_t r = rm[i];
            const float mi = pm[i], ii = pi[i];
            for (uint32_t j = i; !r && j-- > 0 && pm[j] == mi;) r = rm[j] && pi[j] == ii;
            for (uint32_t j = i + 1; !r && j < P && pm[j] == mi; j++) r = rm[j] && pi[j] == ii;
            rm2[i] = r;
        }
        __syncthreads();
        uint32_t newP = 0;
        for (uint32_t bs = 0; bs < P; bs += WAVE) {
            const uint32_t i = bs + lane;
            const bool keep = i < P && !rm2[i];
            const float mi = i < P ? pm[i] : 0.f, ii = i < P ? pi[i] : 0.f;
            const uint64_t km = __ballot(keep);
            const uint32_t pos = newP + (uint32_t)__popcll(km & ((1ull << lane) - 1ull));
            __syncthreads();
            if (keep) { pm[pos] = mi; pi[pos] = ii; }
            newP += (uint32_t)__popcll(km);
            __syncthreads();
        }
        P = newP;
        float t = 0.0f;  // total_ion_current = intensities.iter().sum::<f32>(), scoring.rs:643
        if (lane == 0) for (uint32_t i = 0; i < P; i++) t += pi[i];
        tic = __shfl(t, 0, 64);
        __syncthreads();
    }
    if (lane == 0) out_count[spec] = n_emitted;
}

}  // namespace

size_t prelim_lds_bytes(const DevScorer& sc, const DevBatchView& b, bool wide) {
    const bool fold = sc.min_isotope_err != sc.max_isotope_err;
    size_t n = (size_t)sc.list_cap * (fold ? 16 : 8) + (size_t)sc.kmax * 8 + (size_t)b.fzcap * b.pcap * 8;
    if (!wide) n += ((size_t)sc.wcap / 2 + 1) * 4;
    return (n + 15) & ~(size_t)15;
}
uint32_t rescore_item_cap(const DevBatchView& b, uint32_t max_ions) {
    // one candidate's items must always fit: (ions of the longest peptide) x (fragment charges)
    const uint32_t need = max_ions * (b.fzcap ? b.fzcap : 1);
    return need > 1024 ? need : 1024;
}
size_t rescore_lds_bytes(const DevScorer&, const DevBatchView& b, uint32_t max_ions) {
    size_t n = 192 * 8 + 128 * 4 + (size_t)b.pcap * 8 + (size_t)rescore_item_cap(b, max_ions) * 6 + (size_t)b.pcap * 2;
    return (n + 15) & ~(size_t)15;
}

void launch_prelim(const DevDbView& db, const DevScorer& sc, const DevBatchView& b, const DevWork& w, void* stream) {
    if (b.n == 0) return;
    hipLaunchKernelGGL(prelim_kernel<false>, dim3(b.n), dim3(64), prelim_lds_bytes(sc, b, false), (hipStream_t)stream,
                       db, sc, b, w);
}
void launch_prelim_wide(const DevDbView& db, const DevScorer& sc, const DevBatchView& b, const DevWork& w, void* stream) {
    if (b.n == 0 || w.wide_blocks == 0) return;
    hipLaunchKernelGGL(prelim_kernel<true>, dim3(w.wide_blocks), dim3(64), prelim_lds_bytes(sc, b, true),
                       (hipStream_t)stream, db, sc, b, w);
}
void launch_prelim_open(const DevDbView& db, const DevScorer& sc, const DevBatchView& b, const DevWork& w, void* stream) {
    if (b.n == 0 || w.open_blocks == 0) return;
    const bool fold = sc.min_isotope_err != sc.max_isotope_err;
    const size_t lds = ((size_t)sc.list_cap * (fold ? 16 : 8) + 15) & ~(size_t)15;
    hipLaunchKernelGGL(prelim_open_kernel, dim3(w.open_blocks), dim3(64), lds, (hipStream_t)stream, db, sc, b, w);
}
void launch_rescore(const DevDbView& db, const DevScorer& sc, const DevBatchView& b, const DevWork& w,
                    const double* lnfact_table, uint32_t lnfact_n, uint32_t max_ions, SageFeature* out,
                    uint32_t* out_count, void* stream) {
    if (b.n == 0) return;
    hipLaunchKernelGGL(rescore_kernel, dim3(b.n), dim3(64), rescore_lds_bytes(sc, b, max_ions), (hipStream_t)stream, db,
                       sc, b, w, lnfact_table, lnfact_n, rescore_item_cap(b, max_ions), out, out_count);
}

}  // namespace sagehip
