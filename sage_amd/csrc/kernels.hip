// kernels.hip — gfx950 kernels of the search-and-score path.
//
//   prelim_kernel  : Scorer::initial_hits (scoring.rs:418-462) = precursor-window query
//                    (database.rs:402-425) + matched-fragment counting (scoring.rs:358-375 over
//                    database.rs:480-536) + the nested trim_hits k-selects (scoring.rs:322-329).
//   rescore_kernel : Scorer::build_features / score_candidate / score_chimera_fast
//                    (scoring.rs:478-595, 675-767, 648-672, 598-644).
//
//   tile_*_kernel  : the same for precursor windows too large for one wavefront's LDS counters
//                    (open search, wide-window / DIA): count -> select / replay -> assemble.
//
// One 64-lane wavefront owns one spectrum in the narrow kernels (peaks, fragment-tolerance windows and
// u16 candidate counters in LDS; the window's fragments come either as one contiguous range of the
// peptide-major index or as short runs of a small-tile copy found through a position table); a
// 512-thread workgroup owns one spectrum at a time in the count kernel.  This is sparse
// gather/compare/accumulate work: no MFMA.  Compile with -ffp-contract=off.  DESIGN.md §4 has the rationale
// and the measurements behind each choice.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "crlog.h"
#include "device_types.h"

using namespace sagecore;

namespace sagehip {

namespace {

constexpr uint32_t WAVE = 64;
// LDS words the in-line tie replay of rescore_spectrum may overwrite: the peak bitmap and the peak table behind it (both dead by
// then).  A row of window counts holds (potential + 1) / 2 words, potential <= wcap: capi.hip switches fast ties off for a wcap
// whose rows would not fit, and the kernel checks every row again.
constexpr uint32_t FAST_TIE_WORDS = PBM_WORDS + PLUT_BINS;
constexpr uint32_t CNT_ROW_HEADER = 4;  // DevWork::cnt_store: words in front of a row's counts (keeps them 16-byte aligned)
#ifndef SAGE_PROBE_PER_LANE
#define SAGE_PROBE_PER_LANE 2   // windows whose table reads a lane of the probe kernel keeps in flight (x 64 lanes = one batch); 2 measured best on C3 (LDS footprint vs loads in flight)
#endif
constexpr uint32_t PROBE_BATCH_WORDS = SAGE_PROBE_PER_LANE * 64;
// the run table of a batch: run first / end entry, window lo / hi (PROBE_BATCH_WORDS words each) and the first cell of every run, padded
// with sentinels to a power of two (the owner search halves it)
constexpr uint32_t PROBE_TCS_WORDS = PROBE_BATCH_WORDS <= 64 ? 64 : PROBE_BATCH_WORDS <= 128 ? 128 : PROBE_BATCH_WORDS <= 256 ? 256 : 512;
constexpr uint32_t PROBE_TABLE_WORDS = 4 * PROBE_BATCH_WORDS + PROBE_TCS_WORDS;

// optional per-phase cycle accounting (DevWork::dbg != null): every work item of a launch adds its clock deltas to slot
// [item mod DBG_BLOCKS][kernel*8 + phase] (debug builds of the numbers only; atomics, so slightly perturbing);
// kernel 0 = narrow preliminary, 1 = rescoring, 2 = large-window count, 3 = large-window replay
constexpr uint32_t DBG_BLOCKS = 4096;
// Slots 26..31 of a row count the BYTES the kernels ask memory for (profiling runs only; bench.py reports them next to the
// reference algorithm's bytes): 26 narrow kernel table words, 27 narrow kernel index cells, 28 rescoring ion masses + peaks,
// 29 large-window table words, 30 large-window index cells, 31 candidate words written to the arena.
enum { DBG_NARROW_LUT = 26, DBG_NARROW_CELLS = 27, DBG_RESCORE = 28, DBG_TILE_LUT = 29, DBG_TILE_CELLS = 30, DBG_TILE_CAND = 31 };
struct PhaseClock {
    unsigned long long* slot;
    long long t;
    __device__ __forceinline__ void bytes(int which, unsigned long long n) {  // call from ONE lane
        if (slot) atomicAdd(&slot[which - row_base], n);
    }
    int row_base;
    __device__ __forceinline__ void start(unsigned long long* dbg, uint32_t blk, uint32_t kernel) {
        slot = dbg ? dbg + (size_t)(blk % DBG_BLOCKS) * 32 + kernel * 8 : nullptr;
        row_base = (int)kernel * 8;
        if (slot) t = clock64();
    }
    __device__ __forceinline__ void mark(int phase) {
        if (slot) {
            const long long n = clock64();
            if ((threadIdx.x & 63u) == 0) atomicAdd(&slot[phase], (unsigned long long)(n - t));
            t = n;
        }
    }
    __device__ __forceinline__ void rebase(uint32_t kernel) {  // go on accounting under another kernel's row
        if (slot) slot += (int)kernel * 8 - row_base;
        row_base = (int)kernel * 8;
    }
};

// the same interface compiled to nothing: what the production instances of the per-spectrum kernels carry (a run-time
// "profiling off" check costs scalar registers and branches in kernels that are short of both)
struct NoClock {
    static constexpr unsigned long long* slot = nullptr;
    __device__ __forceinline__ void bytes(int, unsigned long long) {}
    __device__ __forceinline__ void start(unsigned long long*, uint32_t, uint32_t) {}
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void rebase(uint32_t) {}
};

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// Staggered start (experiment, -DSAGE_STAGGER_NS=<ns per slot>): the first generation of a per-spectrum kernel's wavefronts — five
// per SIMD, all launched within microseconds — runs its phases in lockstep (every wavefront waits on memory, then every one wants
// the vector ALU), which is part of what a "cold start" costs; workgroup b of the first 5 x 1024 is held back by (b / 1024) slots.
#ifndef SAGE_STAGGER_NS
#define SAGE_STAGGER_NS 0
#endif
__device__ __forceinline__ void staggered_start(uint32_t blk) {
    if (SAGE_STAGGER_NS > 0 && blk >= 1024u && blk < 5u * 1024u) {
        const uint32_t slot = blk / 1024u;
        const long long until = (long long)__builtin_amdgcn_s_memtime() + (long long)slot * (SAGE_STAGGER_NS * 2);  // (s_memtime counts shader clocks here: ~2 per ns)
        while ((long long)__builtin_amdgcn_s_memtime() < until) __builtin_amdgcn_s_sleep(32);
    }
}

// XCD-aware schedule position.  Workgroup b of a launch is observed to run on XCD b % 8 (MI355X_MICROARCH.md, "Workgroup
// dispatch"; a speed matter only, nothing here depends on it), each XCD with a private 4 MiB L2.  The per-spectrum kernels walk
// the batch in precursor-mass order so that neighbouring wavefronts read overlapping index ranges; dealt round-robin, all eight
// L2s would see the whole mass front (a dozen index tiles: more than one L2 holds).  With this bijective remap the schedule is
// cut into chunks of `chunk` consecutive positions, chunk c goes to XCD c % 8 and every XCD walks its chunks in order: one L2
// serves one narrow mass front, and every XCD still gets every mass range (heavy precursors cost more: contiguous eighths
// would leave the last XCD with the long tail).  The ragged end (fewer than 8 chunks) stays round-robin.
constexpr uint32_t N_XCD = 8;
__device__ __forceinline__ uint32_t xcd_position_fwd(uint32_t b, uint32_t n, uint32_t chunk) {
    if (chunk == 0) return b;
    const uint32_t group = N_XCD * chunk;
    if (b >= n / group * group) return b;
    const uint32_t x = b % N_XCD, i = b / N_XCD;  // the i-th workgroup of XCD x
    return ((i / chunk) * N_XCD + x) * chunk + i % chunk;
}
// (not recursive: a self-call cannot be inlined, and one real call in a kernel costs it the calling convention — a stack
// pointer, callee-saved registers around the call — in kernels that are short of scalar registers)
__device__ __forceinline__ uint32_t xcd_position(uint32_t b, uint32_t n, uint32_t chunk) {
    const uint32_t p = xcd_position_fwd(b, n, chunk & 0x7FFFFFFFu);
    return (chunk & 0x80000000u) ? n - 1 - p : p;  // bit 31: heaviest precursors first
}

// wave-uniform values that come out of memory land in VGPRs; these move them to SGPRs (the value must be uniform)
// A zero register made on the spot: a literal 0 that feeds a store inside a loop is hoisted out of the loop and — in a kernel that is
// out of registers — SPILLED there; it comes back as a scratch load in front of the store, and the s_waitcnt vmcnt(0) behind that load
// also waits for every index cell the kernel has in flight (tile_count_body's clears; rescore_kernel: fresh_zero4).
__device__ __forceinline__ uint32_t fresh_zero() {
    uint32_t z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return z;
}
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ float unif(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ uint64_t uni64(uint64_t v) { return ((uint64_t)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v); }

// wave64 sum through DPP row operations (no LDS crossbar round trips); the total is broadcast from lane 63
__device__ __forceinline__ uint32_t wave_sum_dpp(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1, 0xf, 0xf, false);   // quad_perm:[1,0,3,2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4e, 0xf, 0xf, false);   // quad_perm:[2,3,0,1]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false);  // row_ror:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);  // row_ror:8 -> every lane holds its row's sum
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xf, 0xf, false);  // row_bcast:15 (rows without a source add 0)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xf, 0xf, false);  // row_bcast:31 -> lane 63 holds the total
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// wave64 inclusive prefix sum through DPP row operations (no LDS crossbar round trips): Hillis-Steele inside the rows of 16,
// then the last lane of a row / of the lower half is broadcast into the rows above
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return v;
}

// wave64 inclusive prefix MAXIMUM of unsigned values (identity 0): the same DPP pattern
__device__ __forceinline__ uint32_t wave_incl_scan_max_dpp(uint32_t v) {
#define SAGE_SMAX(CTRL, ROWS, BC)                                                                  \
    {                                                                                              \
        const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWS, 0xf, BC);  \
        v = t > v ? t : v;                                                                         \
    }
    SAGE_SMAX(0x111, 0xf, true)   // row_shr:1
    SAGE_SMAX(0x112, 0xf, true)   // row_shr:2
    SAGE_SMAX(0x114, 0xf, true)   // row_shr:4
    SAGE_SMAX(0x118, 0xf, true)   // row_shr:8
    SAGE_SMAX(0x142, 0xa, false)  // row_bcast:15 into rows 1 and 3
    SAGE_SMAX(0x143, 0xc, false)  // row_bcast:31 into rows 2 and 3
#undef SAGE_SMAX
    return v;
}
// (the total = the scan's last lane: six DPP adds and one v_readlane instead of six ds_bpermute round trips through the LDS crossbar)
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_dpp(v), 63); }

// partition_point over sorted a[lo..hi) of key(a[i]) < bound (STRICT) or <= bound, all 64 lanes
// cooperating: 64 pivots per round (log_65 instead of log_2 dependent loads).
template <bool STRICT>
__device__ __forceinline__ uint32_t wave_partition_point(const float* __restrict__ a, uint32_t lo, uint32_t hi,
                                                         int32_t bound) {
    const uint32_t lane = lane_id();
    while (hi - lo > WAVE) {
        const uint32_t span = hi - lo;
        // 65 pieces of `step` elements with 65 * step - 1 >= span: when all 64 pivots compare true the answer lies in the LAST piece,
        // which must reach `hi` (round 4: ceil(span / 65) left it one element short whenever span was a multiple of 65 — never seen
        // from the full array, at once from the 65-entry brackets of the peptide-mass table)
        const uint32_t step = wpp_step(span);
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

// ---- candidate counters: u16 pairs in LDS ------------------------------------------------------
struct Counters {
    uint32_t* p;
    __device__ __forceinline__ void zero(uint32_t n, uint32_t lane) {
        for (uint32_t i = lane; i < (n + 1) / 2; i += WAVE) p[i] = 0;
    }
    __device__ __forceinline__ void add(uint32_t idx, uint32_t c) { atomicAdd(&p[idx >> 1], c << ((idx & 1) * 16)); }
    __device__ __forceinline__ uint32_t get(uint32_t idx) const { return (p[idx >> 1] >> ((idx & 1) * 16)) & 0xFFFFu; }
};

// LDS written by some lanes of a wavefront and read by others of the SAME wavefront: DS operations of one
// wavefront complete in order, so only the compiler has to be kept from reordering / caching across this point.
// ... and where global loads are meant to STAY in flight across the point (rescore_spectrum builds its LDS tables while the
// gathers of the candidates' records are still on their way): wait for the wavefront's own LDS operations only.  One wavefront
// per workgroup in every kernel that uses it.
__device__ __forceinline__ void lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding global load and
// store of the wavefront (vmcnt(0)), which would serialise loads issued ahead of a barrier on purpose.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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
__device__ __forceinline__ float lane_valuef(float v, uint32_t src_lane) {
    src_lane = __builtin_amdgcn_readfirstlane(src_lane);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (int)src_lane));
}
// sift_down (heap.rs:40-60) of value `moving` placed at `index`.  sift_down always descends to the smaller child
// (the left one on a tie), a path that does not depend on the value being sifted: every lane compares ITS two
// children once (two cross-lane reads), a ballot turns that into one bit per node, the path is then walked with
// scalar bit operations, its values are read with v_readlane, and — values along a heap path never decrease —
// `moving` stops at the first path value that is not smaller.  All control flow is wave-uniform.
__device__ __forceinline__ void wh_sift_from(WaveHeap& h, uint32_t len, uint32_t index, uint64_t moving) {
    const uint32_t lane = lane_id();
    const uint32_t l = 2 * lane + 1, r = l + 1;
    const uint32_t lo_l = (uint32_t)__shfl((int)h.lo, (int)(l & 63u), 64), hi_l = (uint32_t)__shfl((int)h.hi, (int)(l & 63u), 64);
    const uint32_t lo_r = (uint32_t)__shfl((int)h.lo, (int)(r & 63u), 64), hi_r = (uint32_t)__shfl((int)h.hi, (int)(r & 63u), 64);
    const uint64_t vl = ((uint64_t)hi_l << 32) | lo_l, vr = ((uint64_t)hi_r << 32) | lo_r;
    const uint64_t rightmin = __ballot(r < len && vr < vl);  // bit p: the right child of node p is strictly smaller
    index = __builtin_amdgcn_readfirstlane(index);
    uint32_t path[7];
    uint64_t pv[7];
    uint32_t n = 0;  // nodes below `index` on the path that hold a value smaller than `moving`
    path[0] = index;
    uint32_t dst = index;  // where `moving` ends up
    bool go = true;
#pragma unroll
    for (uint32_t j = 0; j < 6; j++) {
        const uint32_t lc = 2 * path[j] + 1;
        go = go && lc < len;
        path[j + 1] = go ? lc + (uint32_t)((rightmin >> (path[j] & 63u)) & 1ull) : 0;
        pv[j + 1] = go ? wh_get(h, path[j + 1]) : ~0ull;
        go = go && pv[j + 1] < moving;
        n += go ? 1u : 0u;
        dst = go ? path[j + 1] : dst;
    }
    // shift the path up by one level and drop `moving` where it stopped (slice.swap at every level)
#pragma unroll
    for (uint32_t j = 0; j < 6; j++)
        if (j < n) wh_set(h, path[j], pv[j + 1]);
    wh_set(h, dst, moving);
}
__device__ __forceinline__ void wh_build(WaveHeap& h, uint32_t k) {  // heap.rs:13-15
    for (uint32_t i = k / 2; i-- > 0;) wh_sift_from(h, k, i, wh_get(h, i));
}
__device__ __forceinline__ void wh_offer(WaveHeap& h, uint32_t k, uint64_t v) {  // heap.rs:21-27
    if (k && v > wh_get(h, 0)) wh_sift_from(h, k, 0, v);  // slice.swap(i, 0): the displaced minimum is truncated away
}

// The same heap with 32-bit keys, for the k-select of ONE precursor-window query: charge and isotope error are constant inside a
// query, so PreScore's order is (matched, peptide) = (matched, candidate slot), and `matched << S | slot` orders identically (an
// empty slot is key 0).  This replay is THE serial chain of the exact retry pass — a large-window query feeds it thousands of
// offers, one after the other, in a wavefront that runs almost alone on its SIMD — and such a wavefront retires an instruction
// every ~7 cycles whatever the instruction: what counts is how many there are.  So: plain sift_down, scalar control flow, two
// v_readlane per level for the children, one v_writelane per level; no cross-lane permutes, no ballots, no per-node copies
// (three earlier versions that shortened the dependency chain instead executed 2-3 x the instructions and were slower;
// profiles/r03_replay.md).
struct Heap32 {
    uint32_t h;  // node `lane`
};
__device__ __forceinline__ uint32_t wh32_get(const Heap32& H, uint32_t idx) {
    return (uint32_t)__builtin_amdgcn_readlane((int)H.h, (int)__builtin_amdgcn_readfirstlane(idx));
}
// vdst[lane `sel`] = val, both wave-uniform: one v_writelane_b32 (no builtin for it in this toolchain).  gfx9 lets a VALU
// instruction read ONE scalar register, so the lane select travels in M0; the compiler's hazard recogniser does not look inside
// an asm statement, hence the wait states behind the M0 write are spelled out.
__device__ __forceinline__ void writelane(uint32_t& vdst, uint32_t val, uint32_t sel) {
    // (M0 is a reserved register: the compiler never keeps a value in it across statements, it loads it right in front of the
    // few instructions that read it — so it is not, and cannot be, listed as clobbered)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 1\n\tv_writelane_b32 %0, %1, m0" : "+v"(vdst) : "s"(val), "s"(sel));
}
__device__ __forceinline__ void wh32_init(Heap32& H, uint32_t keys, uint32_t k) { H.h = lane_id() < k ? keys : 0xFFFFFFFFu; }
// sift_down (heap.rs:40-60) of value `moving` placed at `index`: descend to the smaller child (the left one on a tie) while it
// is smaller than `moving`, shifting it up (slice.swap at every level).  All control flow is wave-uniform.
__device__ __forceinline__ void wh32_sift_from(Heap32& H, uint32_t len, uint32_t index, uint32_t moving) {
    uint32_t p = __builtin_amdgcn_readfirstlane(index);
    moving = __builtin_amdgcn_readfirstlane(moving);
    for (;;) {
        const uint32_t l = 2 * p + 1;
        if (l >= len) break;
        uint32_t c = l;
        uint32_t cv = (uint32_t)__builtin_amdgcn_readlane((int)H.h, (int)l);
        if (l + 1 < len) {
            const uint32_t rv = (uint32_t)__builtin_amdgcn_readlane((int)H.h, (int)(l + 1));
            if (rv < cv) { cv = rv; c = l + 1; }
        }
        if (!(cv < moving)) break;
        writelane(H.h, cv, p);
        p = c;
    }
    writelane(H.h, moving, p);
}
__device__ __forceinline__ void wh32_build(Heap32& H, uint32_t k) {  // heap.rs:13-15
    for (uint32_t i = k / 2; i-- > 0;) wh32_sift_from(H, k, i, wh32_get(H, i));
}
__device__ __forceinline__ void wh32_offer(Heap32& H, uint32_t k, uint32_t v) {  // heap.rs:21-27
    if (k && v > wh32_get(H, 0)) wh32_sift_from(H, k, 0, v);
}

// heap.rs:24-25 + :40-60 for a value that enters at the root, ALL LANES AT ONCE (round 6: the replay of a large-window query is
// one serial chain of thousands of these — the longest query of a retry pass IS the pass's duration — and the loop above spends
// ~1 200 cycles per offer on six levels of readlane / compare / writelane that depend on each other).  sift_down always descends
// to the smaller child, the left one on a tie (heap.rs:43-51): which child that is does not depend on the value being sifted, so
// every node looks at its two children once (two ds_bpermute, in flight together), one ballot gives the "right child is
// strictly smaller" bit of every node, the root-to-leaf path follows from bit operations on scalars, and — values along a heap
// path never decrease — the new value stops behind the path nodes that are smaller (a prefix of the path: one more ballot).  A path
// node above the stop takes its smaller child's value (it has it already), the node at the stop takes `v`; a node's depth is a
// function of its lane.  Same array as the loop, every time (the GPU suite's exact paths and the config-scale parity run through it).
// (round 6, second step: the path was a scalar loop, ~12 dependent scalar instructions for each of six levels — 43 scalar instructions
// per offer, the replay's time.  Which nodes lie on the path is a question every lane answers for itself: lane L is on it iff every
// ancestor of L chose the child that leads to L, i.e. `rm` restricted to L's ancestors equals a constant of the lane — two 64-bit
// masks per lane, made once per kernel: HeapPath.)
struct HeapPath {
    uint64_t anc, want;  // bit a: node a is an ancestor of this lane's node / ... and the way down to it takes a's RIGHT child
    uint32_t depth;
};
__device__ __forceinline__ HeapPath heap_path_of_lane() {
    HeapPath hp{0ull, 0ull, 0u};
    const uint32_t n = lane_id() + 1u;  // (1-based: the children of m are 2m and 2m + 1)
    for (uint32_t c = n; c > 1u; c >>= 1) {
        const uint32_t a = (c >> 1) - 1u;
        hp.anc |= 1ull << a;
        if (c & 1u) hp.want |= 1ull << a;
    }
    hp.depth = 31u - (uint32_t)__clz((int)n);
    return hp;
}
__device__ __forceinline__ void wh32_replace_root(Heap32& H, uint32_t k, uint32_t v, const HeapPath& path) {
    const uint32_t lane = lane_id();
    const uint32_t l = 2 * lane + 1, r = l + 1;
    const uint32_t hl = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((l & 63u) << 2), (int)H.h);
    const uint32_t hr = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((r & 63u) << 2), (int)H.h);
    const bool right = r < k && hr < hl;
    const uint64_t rm = __ballot(right);
    const bool onpath = lane < k && (rm & path.anc) == path.want;
    const uint32_t stop = (uint32_t)__popcll(__ballot(onpath && lane != 0u && H.h < v));  // path nodes below the root that move up
    if (onpath) H.h = path.depth < stop ? (right ? hr : hl) : path.depth == stop ? v : H.h;
}
__device__ __forceinline__ void wh32_offer_par(Heap32& H, uint32_t k, uint32_t v, const HeapPath& path) {  // heap.rs:21-27
    if (k && v > wh32_get(H, 0)) wh32_replace_root(H, k, v, path);
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
    const uint32_t lit = (uint32_t)(n < room ? n : room);  // <= kmax (one trip unless report_psms > 32)
    for (uint32_t done = 0; done < lit; done += WAVE) ulist_append(c, PRESCORE_EMPTY, lit - done < WAVE ? lit - done : WAVE, kmax);
    c.len += n - lit;
}
// trim_hits (scoring.rs:322-329); called by one whole wavefront (the list lives in LDS).
// exact: replay bounded_min_heapify, the list ends up in the reference's heap layout.  !exact: keep the same k entries
// (the k largest; equal entries are interchangeable) in descending order, by ranking every stored entry.
__device__ __forceinline__ void ulist_trim(UList& c, uint32_t report_psms, bool exact) {
    const uint32_t lane = lane_id();
    const uint32_t k = trim_k(c.len, report_psms);
    if (c.len > k && exact) {
        wave_sync();
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
        wave_sync();
        if (lane < k) c.items[lane] = ((uint64_t)h.hi << 32) | h.lo;
        wave_sync();
    } else if (c.len > k) {
        // (entries of the logical list that are not stored are PreScore::default(): the smallest value there is)
        wave_sync();
        uint64_t keep_v[4];   // list_cap <= 4 * 64 entries (capi.hip sizes it; larger lists take the exact path)
        uint32_t keep_r[4];
        const uint32_t nst = c.stored;
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
            const uint32_t i = q * WAVE + lane;
            keep_v[q] = i < nst ? c.items[i] : 0ull;
            uint32_t rank = 0;
            if (i < nst)
                for (uint32_t j = 0; j < nst; j++) {
                    const uint64_t o = c.items[j];  // (same address in every lane: an LDS broadcast)
                    rank += (o > keep_v[q]) || (o == keep_v[q] && j < i);
                }
            keep_r[q] = i < nst ? rank : 0xFFFFFFFFu;
        }
        wave_sync();
#pragma unroll
        for (uint32_t q = 0; q < 4; q++)
            if (keep_r[q] < k) c.items[keep_r[q]] = keep_v[q];
        for (uint32_t i = nst + lane; i < k; i += WAVE) c.items[i] = PRESCORE_EMPTY;  // fewer stored than k: defaults fill up
        wave_sync();
    }
    c.stored = k;
    c.len = k;
}

// ---- k-select wider than a wavefront: report_psms > 32, k = max(50, 2 * report_psms) up to BIG_K = 1024 (BIGK instantiations) ------
// bounded_min_heapify (heap.rs:7-60) with the heap in LDS, replayed by lane 0 with core.h's sift_down (the host's, the oracle's);
// the wavefront only skims: 64 offers at a time are tested against the current minimum — the minimum never decreases, so an offer
// that fails now would fail later — and the survivors are handed to lane 0 in order.  Slow next to the register heaps above and
// meant to be: a search that reports more than 32 PSMs per spectrum is a rare configuration, it has to be RIGHT.
__device__ __forceinline__ void lh_build(uint64_t* a, uint32_t k) {
    wave_sync();
    if (lane_id() == 0) heap_build(a, k);
    wave_sync();
}
// lane i offers v (has: it has one), lanes in the list's order
__device__ __forceinline__ void lh_offer_batch(uint64_t* a, uint32_t k, uint64_t v, bool has) {
    if (!k) return;
    uint64_t mask = __ballot(has && v > a[0]);
    if (!mask) return;
    while (mask) {
        const uint32_t bit = (uint32_t)__ffsll((long long)mask) - 1;
        mask &= mask - 1;
        const uint64_t o = lane_value(v, bit);
        if (lane_id() == 0) heap_offer(a, k, o);
    }
    wave_sync();
}
// trim_hits (scoring.rs:322-329) of a list in LDS, in place as the reference does it (the heap is the slice's first k entries)
__device__ __forceinline__ void ulist_trim_big(UList& c, uint32_t report_psms) {
    const uint32_t lane = lane_id();
    const uint32_t k = trim_k(c.len, report_psms);
    if (c.len > k) {
        lh_build(c.items, k);
        for (uint32_t base = k; base < c.stored; base += WAVE) {
            const bool has = base + lane < c.stored;
            lh_offer_batch(c.items, k, has ? c.items[base + lane] : 0ull, has);
        }
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
    uint32_t* ptab;  // probe variant: run table of one batch of windows, PROBE_TABLE_WORDS words
    uint32_t* cnt;
};

// `big` (the instances for lists wider than a wavefront only; null: LDS): the lists and the heap in a global-memory workspace of the
// workgroup (DevWork::hugebuf) — preliminary lists that no longer fit a compute unit's LDS (report_psms in the hundreds x many
// precursor-window queries per spectrum; report_psms beyond 512)
__device__ __forceinline__ PrelimLds carve_prelim(unsigned char* smem, const DevScorer& sc, const DevBatchView& b, unsigned char* big = nullptr) {
    PrelimLds l;
    size_t off = 0;
    const bool fold = sc.min_isotope_err != sc.max_isotope_err;
    if (big) {
        l.listA = (uint64_t*)big;
        l.listB = l.listA + (fold ? (size_t)sc.list_cap : 0);
        l.heap = l.listB + sc.list_cap;
    } else {
        l.listA = (uint64_t*)(smem + off); off += fold ? (size_t)sc.list_cap * 8 : 0;
        l.listB = (uint64_t*)(smem + off); off += (size_t)sc.list_cap * 8;
        l.heap = (uint64_t*)(smem + off); off += (size_t)sc.kmax * 8;
    }
    // probe variant: only the peak masses are staged (win_lo[0..pcap)); each lane derives its window bounds on the fly
    const size_t win = b.probe ? (size_t)b.pcap * 4 : (size_t)b.fzcap * b.pcap * 4;
    l.win_lo = (float*)(smem + off); off += win;
    l.win_hi = (float*)(smem + off); off += b.probe ? 0 : win;
    off = (off + 3) & ~(size_t)3;
    l.ptab = (uint32_t*)(smem + off); off += b.probe ? (size_t)PROBE_TABLE_WORDS * 4 : 0;
    l.cnt = (uint32_t*)(smem + off);
    return l;
}

__host__ __device__ inline size_t prelim_big_bytes(const DevScorer& sc) {  // the lists and the heap (carve_prelim)
    const bool fold = sc.min_isotope_err != sc.max_isotope_err;
    return (size_t)sc.list_cap * (fold ? 16 : 8) + (size_t)sc.kmax * 8;
}
__host__ __device__ inline size_t prelim_layout_bytes(const DevScorer& sc, const DevBatchView& b, bool huge = false) {
    size_t n = (huge ? 0 : prelim_big_bytes(sc)) +
               (b.probe ? (size_t)b.pcap * 4 + 4 + (size_t)PROBE_TABLE_WORDS * 4 : (size_t)b.fzcap * b.pcap * 8);
    n += ((size_t)sc.wcap / 2 + 1) * 4;
    return (n + 15) & ~(size_t)15;
}

// ---- shared by both preliminary kernels ----------------------------------------------------------
struct SpecInfo {
    uint64_t p0;
    uint32_t P, z0, z1, nfz_max;
    float mzp;
    Tol iso_tol;
};
__device__ __forceinline__ SpecInfo load_spec(const DevScorer& sc, const DevBatchView& b, uint32_t spec) {
    SpecInfo s;
    s.p0 = uni64(b.peak_off[spec]);
    s.P = (uint32_t)(uni64(b.peak_off[spec + 1]) - s.p0);
    const uint32_t zraw = uni(b.precursor_charge[spec]);
    if (sc.wide_window || zraw == 0 || sc.override_precursor_charge) {  // scoring.rs:423, 437, 442
        s.z0 = sc.min_precursor_charge;
        s.z1 = sc.max_precursor_charge;
    } else {
        s.z0 = s.z1 = zraw;
    }
    s.nfz_max = 0;
    for (uint32_t z = s.z0; z <= s.z1; z++) {
        const uint32_t m = max_fragment_charge(sc.max_fragment_charge, z) - 1;
        s.nfz_max = m > s.nfz_max ? m : s.nfz_max;
    }
    if (s.nfz_max > b.fzcap) s.nfz_max = b.fzcap;  // (upload sized fzcap from the same rule)
    s.mzp = unif(b.precursor_mz[spec]) - PROTON;  // scoring.rs:420
    s.iso_tol.kind = 2;
    s.iso_tol.lo = -2.4f;
    s.iso_tol.hi = 2.4f;  // scoring.rs:430
    if (b.isolation_lo && b.isolation_hi) {
        const float a = unif(b.isolation_lo[spec]), c = unif(b.isolation_hi[spec]);
        if (a == a && c == c) { s.iso_tol.lo = a; s.iso_tol.hi = c; }
    }
    return s;
}

// The same from the batch's schedule records (DevBatchView::sched): record `pos` holds what load_spec reads of spectrum order[pos] — one
// trip, to a line the neighbouring blocks of the XCD share, instead of order[pos] and then five reads at random places.
__device__ __forceinline__ SpecInfo load_spec_sched(const DevScorer& sc, const DevBatchView& b, uint32_t pos, uint32_t& spec) {
    const uint4 r0 = b.sched[2 * (size_t)pos], r1 = b.sched[2 * (size_t)pos + 1];
    spec = uni(r0.x);
    SpecInfo s;
    s.p0 = ((uint64_t)uni(r0.w) << 32) | uni(r0.z);
    s.P = uni(r0.y);
    const uint32_t zraw = uni(r1.x);
    if (sc.wide_window || zraw == 0 || sc.override_precursor_charge) {  // scoring.rs:423, 437, 442
        s.z0 = sc.min_precursor_charge;
        s.z1 = sc.max_precursor_charge;
    } else {
        s.z0 = s.z1 = zraw;
    }
    s.nfz_max = 0;
    for (uint32_t z = s.z0; z <= s.z1; z++) {
        const uint32_t m = max_fragment_charge(sc.max_fragment_charge, z) - 1;
        s.nfz_max = m > s.nfz_max ? m : s.nfz_max;
    }
    if (s.nfz_max > b.fzcap) s.nfz_max = b.fzcap;
    s.mzp = unif(__uint_as_float(r1.y)) - PROTON;  // scoring.rs:420
    s.iso_tol.kind = 2;
    s.iso_tol.lo = -2.4f;
    s.iso_tol.hi = 2.4f;  // scoring.rs:430
    const float a = unif(__uint_as_float(r1.z)), c = unif(__uint_as_float(r1.w));  // (NaN bits where the batch has no isolation windows)
    if (a == a && c == c) { s.iso_tol.lo = a; s.iso_tol.hi = c; }
    return s;
}

// IndexedDatabase::query (database.rs:402-425) by one wavefront: [left, right] candidate slots and the
// [first, end) peptide range after the edge rule of database.rs:526-531
struct Window {
    uint32_t left, right, first, end;
};
// `lut` (DevDbView::pep_lut, may be null): the table of pep_mono's partition points at multiples of 1 / inv_w.  Both bounds of an
// ordinary window (0 <= plo <= phi inside the table) then lie in the one or two bins the table brackets them with: two scalar
// reads and one wave-wide read each, the same partition points as the full search finds (which everything else still takes).
template <bool GALLOP>
__device__ __forceinline__ Window query_window(const float* __restrict__ pep_mono, const uint32_t np, const Tol& ptol, float center,
                                               const uint32_t* __restrict__ lut = nullptr, const uint32_t bins = 0, const float inv_w = 0.0f) {
    float plo, phi;
    tol_bounds(ptol, center, plo, phi);
    Window q;
    uint32_t left, right;
    if (lut && plo >= 0.0f && phi >= plo && phi * inv_w < (float)bins) {
        const uint32_t b0 = uni((uint32_t)(plo * inv_w)), b1 = uni((uint32_t)(phi * inv_w));  // (the scaling is exact: floor)
        const uint32_t l0 = lut[b0], l1 = lut[b0 + 1], r0 = lut[b1], r1 = lut[b1 + 1];
#ifndef SAGE_QUERY_ONE_TRIP
#define SAGE_QUERY_ONE_TRIP 1
#endif
        if (SAGE_QUERY_ONE_TRIP && l1 - l0 < WAVE && r1 - r0 < WAVE) {
            // Both brackets fit a wavefront (the usual case: a bin of 1/128 Da): ONE trip to pep_mono instead of four in a row — the
            // two partition points from two reads in flight together, and the edge rule's two masses (below) out of the same
            // registers: lanes of `vl` hold [l0 - 1, l0 + 63), lanes of `vr` [r0, r0 + 64); `left` is l0 - 1 .. l1 - 1 and `right`
            // r0 .. r1, both inside.  Same comparisons on the same values as the searches of the other branch.
            const uint32_t lane = lane_id();
            const uint32_t base_l = l0 ? l0 - 1 : 0;
            const uint32_t il = base_l + lane, ir = r0 + lane;
            const float vl = il < np ? pep_mono[il] : 0.0f, vr = ir < np ? pep_mono[ir] : 0.0f;
            const uint32_t ppl = l0 + (uint32_t)__popcll(__ballot(il >= l0 && il < l1 && order_key(vl) < order_key(plo)));
            const uint32_t ppr = r0 + (uint32_t)__popcll(__ballot(ir < r1 && order_key(vr) <= order_key(phi)));
            left = ppl ? ppl - 1 : 0;
            // (the search of the other branch starts at max(r0, left): everything in front of `left` is below plo <= phi, so its
            // result is the full bracket's, or `left` where that is larger)
            right = ppr > left ? ppr : left;
            q.left = left;
            q.right = right;
            q.first = left;
            q.end = right;
            const float m_left = lane_valuef(vl, left - base_l);
            const float m_right = right == left ? m_left : lane_valuef(vr, right - r0);
            if (left < np && !(m_left >= plo)) q.first = left + 1;
            if (right < np && m_right <= phi) q.end = right + 1;
            return q;
        }
        left = wave_partition_point<true>(pep_mono, l0, l1, order_key(plo));
        left = left ? left - 1 : 0;
        right = wave_partition_point<false>(pep_mono, r0 > left ? r0 : left, r1, order_key(phi));
    } else {
        left = wave_partition_point<true>(pep_mono, 0, np, order_key(plo));
        left = left ? left - 1 : 0;
        uint32_t ghi = np;
        if (GALLOP) {  // the window is short in a narrow search: bracket it by galloping from `left` before searching
            for (uint64_t span = WAVE;; span *= 16) {
                ghi = (uint64_t)left + span < np ? (uint32_t)(left + span) : np;
                if (ghi == np || order_key(pep_mono[ghi - 1]) > order_key(phi)) break;
            }
        }
        right = wave_partition_point<false>(pep_mono, left, ghi, order_key(phi));
    }
    q.left = left;
    q.right = right;
    q.first = left;
    q.end = right;
    if (left < np && !(pep_mono[left] >= plo)) q.first = left + 1;
    if (right < np && pep_mono[right] <= phi) q.end = right + 1;
    return q;
}

// trim_hits of one precursor-window query WITHOUT replaying the heap: the same k slots — the k largest by (matched count,
// slot); bounded_min_heapify keeps exactly that set — appended in slot order.  A 64-bin histogram of the counts gives the
// k-th largest count T; every slot above T is kept, and of the slots equal to T the LAST ones (larger peptide index wins
// a tie in PreScore's order).  Returns false (nothing appended) when a count does not fit the histogram.
__device__ __forceinline__ bool fast_select(const PrelimLds& L, const Counters& cnt, uint32_t potential, uint32_t k, uint32_t left,
                                            uint32_t z, int iso, uint32_t kmax, UList& target, uint32_t& scored) {
    const uint32_t lane = lane_id();
    uint32_t* hist = (uint32_t*)L.heap;  // (kmax * 8 >= 400 bytes: 64 bins, then reused for the selected entries)
    wave_sync();
    hist[lane] = 0;
    wave_sync();
    for (uint32_t base = 0; base < potential; base += WAVE) {
        const uint32_t i = base + lane;
        const uint32_t c = i < potential ? cnt.get(i) : 0;
        if (c) atomicAdd(&hist[c < 63 ? c : 63], 1u);
    }
    wave_sync();
    const uint32_t mine = hist[lane];
    // number of slots with count >= lane: the suffix sum, as total - inclusive prefix + own (DPP scan: no crossbar round trips)
    const uint32_t incl_pre = wave_incl_scan_dpp(mine);
    const uint32_t suffix = (uint32_t)__builtin_amdgcn_readlane((int)incl_pre, 63) - incl_pre + mine;
    // (wave-uniform lane indices: v_readlane, not a trip through the LDS crossbar)
    if (__builtin_amdgcn_readlane((int)mine, 63) != 0) return false;  // a count >= 63: keep the exact path
    scored = (uint32_t)__builtin_amdgcn_readlane((int)suffix, 1);     // slots with count > 0
    const uint64_t okm = __ballot(lane >= 1 && suffix >= k);
    const uint32_t T = uni(okm ? 63u - (uint32_t)__clzll((long long)okm) : 0u);  // k-th largest count (0: fewer than k non-empty)
    const uint32_t n_gt = (uint32_t)__builtin_amdgcn_readlane((int)suffix, (int)(T + 1 < 64 ? T + 1 : 63));  // (T <= 62 here)
    const uint32_t n_eq = T ? (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)T) : 0;
    const uint32_t take_eq = T ? k - n_gt : 0, skip_eq = n_eq - take_eq;
    wave_sync();
    uint64_t* sel = L.heap;
    uint32_t nsel = 0, eq_seen = 0;
    for (uint32_t base = 0; base < potential; base += WAVE) {
        const uint32_t i = base + lane;
        const uint32_t c = i < potential ? cnt.get(i) : 0;
        const uint64_t eqm = __ballot(T != 0 && c == T);
        const uint32_t eq_idx = eq_seen + (uint32_t)__popcll(eqm & ((1ull << lane) - 1ull));
        const bool take = c > T || (T != 0 && c == T && eq_idx >= skip_eq);
        const uint64_t tm = __ballot(take);
        if (take) sel[nsel + (uint32_t)__popcll(tm & ((1ull << lane) - 1ull))] = pack_prescore(c, left + i, z, iso);
        nsel += (uint32_t)__popcll(tm);
        eq_seen += (uint32_t)__popcll(eqm);
    }
    wave_sync();
    // nsel == k, or fewer when the window holds fewer than k non-empty slots: PreScore::default() entries fill up
    ulist_append(target, lane < nsel ? sel[lane] : PRESCORE_EMPTY, k, kmax);
    wave_sync();
    return true;
}

// ---- narrow windows: one wavefront per spectrum, counters in LDS -----------------------------------
// Two ways to count the matched fragments of a precursor window, same predicate (database.rs:526-533), same counts:
//   PROBE == false : stream the window's fragments from the peptide-major copy of the index (one contiguous, coalesced
//                    range) and count, per fragment, the experimental windows that contain it (binary searches in LDS).
//                    Cost ~ candidates x fragments per peptide: best for windows of a few dozen candidates.
//   PROBE == true  : one lane per (peak, fragment charge) window looks its run up in the tile-major copy (two table
//                    reads + ~10-20 entries).  Cost ~ peaks x fragment charges, independent of the window: best from
//                    ~100 candidates up.
// The C ABI picks per batch from the mean window size (capi.hip: sage_hip_batch_upload).
// Build-time knobs, with the values measured best on MI355X (scripts/variants.sh, scripts/ab_libs.sh; C3: 26.3 -> 29.8 M
// spectra/s).  Both per-spectrum kernels are bound by dependent memory / LDS round trips, so resident wavefronts matter more
// than registers per wavefront: capping the VGPR budget at 6 waves per SIMD (80 VGPRs, a handful of spills outside the inner
// loops) and keeping LDS per wavefront under 160 KB / 24 buys ~10 %; past 6 waves nothing more comes.
#ifndef SAGE_PRELIM_WAVES
#define SAGE_PRELIM_WAVES 5  // 0: leave the occupancy to the compiler (A/B on C3: 5 > 6 > 7 > 8)
#endif
#ifndef SAGE_RESCORE_WAVES
#define SAGE_RESCORE_WAVES 5  // (A/B on C3 with the cooperative matching: 5 > 6)
#endif
#ifndef SAGE_PROBE_PER_LANE
#define SAGE_PROBE_PER_LANE 2   // windows whose table reads a lane of the probe kernel keeps in flight (x 64 lanes = one batch); 2 measured best on C3 (LDS footprint vs loads in flight)
#endif
#ifndef SAGE_PROBE_CELLS
#define SAGE_PROBE_CELLS 4      // 16-byte index cells a lane keeps in flight per pass over the flattened runs
#endif
constexpr uint32_t PROBE_PER_LANE = SAGE_PROBE_PER_LANE;
constexpr uint32_t PROBE_BATCH = PROBE_PER_LANE * 64;
constexpr uint32_t PROBE_CELLS = SAGE_PROBE_CELLS;
constexpr uint32_t NO_WINDOW = 0xFFFFFFFFu;
static_assert(PROBE_BATCH == PROBE_BATCH_WORDS && PROBE_BATCH <= 512, "the run table's layout (carve_prelim) is the kernel's");
static_assert(PROBE_TCS_WORDS - PROBE_BATCH <= WAVE, "one sentinel per lane pads the run starts");
#ifndef SAGE_NARROW_WAVES
#define SAGE_NARROW_WAVES 5  // the fused narrow kernel
#endif
#if SAGE_NARROW_WAVES
#define SAGE_NARROW_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(SAGE_NARROW_WAVES, SAGE_NARROW_WAVES)))
#else
#define SAGE_NARROW_WAVES_ATTR
#endif
#if SAGE_PRELIM_WAVES
#define SAGE_PRELIM_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(SAGE_PRELIM_WAVES, SAGE_PRELIM_WAVES)))
#else
#define SAGE_PRELIM_WAVES_ATTR
#endif
#if SAGE_RESCORE_WAVES
#define SAGE_RESCORE_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(SAGE_RESCORE_WAVES, SAGE_RESCORE_WAVES)))
#else
#define SAGE_RESCORE_WAVES_ATTR
#endif
// ---- arguments where they are used ---------------------------------------------------------------------------------------------
// The per-spectrum kernels take ~100 scalar registers' worth of arguments (four views) and have 100 scalar registers.  Loaded at
// the kernel's entry — where the compiler puts kernarg loads — most of them are written to spill lanes at once and read back
// phase by phase: v_writelane / v_readlane, vector-ALU issue slots of kernels that are bound by exactly those (round 4:
// ~350 of prelim_kernel's ~2 000 vector instructions per spectrum).  But a kernel's arguments never need saving: they sit in the
// kernarg segment, one SCALAR load away.  ArgRef<K> is how prelim_spectrum reads them:
//   K = void:           from the references it was handed (the kernels whose argument list is something else)
//   K = PrelimKernargs: from the kernarg segment through a pointer the compiler cannot see through (`refresh()` at the head of
//                       every phase) — the loads stay at the head of the phase that uses them, nothing lives across phases.
// (LateArgs, further down, is the same idea for the tail of rescore_spectrum.)
struct PrelimKernargs {  // THE argument list of prelim_kernel (one struct: the segment's layout is this struct's)
    DevDbView db;
    DevScorer sc;
    DevBatchView b;
    DevWork w;
};
#define SAGE_LOAD_TOL(t) Tol{(t).kind, (t).lo, (t).hi}  /* a Tol out of either address space, field by field */
template <class K>
struct ArgRef {  // K = void
    const DevDbView& db_;
    const DevScorer& sc_;
    const DevBatchView& b_;
    __device__ __forceinline__ ArgRef(const DevDbView& d, const DevScorer& s, const DevBatchView& b) : db_(d), sc_(s), b_(b) {}
    __device__ __forceinline__ void refresh() {}
    __device__ __forceinline__ const DevDbView& db() const { return db_; }
    __device__ __forceinline__ const DevScorer& sc() const { return sc_; }
    __device__ __forceinline__ const DevBatchView& b() const { return b_; }
};
template <>
struct ArgRef<PrelimKernargs> {
    typedef const __attribute__((address_space(4))) PrelimKernargs* Segment;
    Segment ka;
    __device__ __forceinline__ ArgRef(const DevDbView&, const DevScorer&, const DevBatchView&)
        : ka((Segment)__builtin_amdgcn_kernarg_segment_ptr()) {}
    __device__ __forceinline__ void refresh() {  // (from the builtin every time: a `ka` carried across branches would end up in a VGPR)
        ka = (Segment)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
    }
    __device__ __forceinline__ const __attribute__((address_space(4))) DevDbView& db() const { return ka->db; }
    __device__ __forceinline__ const __attribute__((address_space(4))) DevScorer& sc() const { return ka->sc; }
    __device__ __forceinline__ const __attribute__((address_space(4))) DevBatchView& b() const { return ka->b; }
};

// Scorer::initial_hits (scoring.rs:418-462) of ONE spectrum by one wavefront: the final preliminary list ends up in L.listB.
// `exact`: every trim_hits replays bounded_min_heapify (the list is in the reference's heap layout); else the order-free trims.
struct PrelimResult {
    uint32_t stored;            // entries of the preliminary list, L.listB[0 .. stored)
    uint32_t matched, scored;   // InitialHits.matched_peaks, .scored_candidates
    bool ok;                    // false: a candidate list overflowed its LDS capacity
    bool deferred;              // some precursor window exceeds the LDS counters: the spectrum belongs to the large-window kernels
    bool untrimmed;             // no trim_hits had anything to drop (every list stayed within its k): the list is the reference's
                                //     Vec as it stands, whatever the trim mode — a tie at a reported rank needs no exact pass
    uint32_t q_left, q_potential;  // the LAST precursor-window query: first candidate slot's peptide and the number of slots; its
                                   //     counts are still in L.cnt when prelim_spectrum returns
};
template <bool PROBE, bool BIGK = false, class KA = void, class PC>
__device__ __forceinline__ PrelimResult prelim_spectrum(const DevDbView& db_, const DevScorer& sc_, const DevBatchView& b_, const PrelimLds& L,
                                                        const SpecInfo& si, const bool exact, PC& pc) {
    const uint32_t lane = lane_id();
    ArgRef<KA> av(db_, sc_, b_);  // (every argument through `av`, refreshed phase by phase: see ArgRef)
    av.refresh();
    Counters cnt;
    cnt.p = L.cnt;
    PrelimResult res{0u, 0u, 0u, true, false, true, 0u, 0u};
    {
        const uint32_t P = si.P, nfz_max = si.nfz_max;
        const float* __restrict__ masses = av.b().masses + si.p0;
        const uint32_t pcap = av.b().pcap;
        const Tol ftol0 = SAGE_LOAD_TOL(av.sc().fragment_tol);

        // PROBE: the peak masses are REQUESTED here and staged behind the spectrum's first precursor-window search, whose chain of
        // dependent reads (position table -> peptide masses -> window edges) does not need them: one round trip less in a row.
        float pk0 = 0.f, pk1 = 0.f, pk2 = 0.f;
        Window q_first{0u, 0u, 0u, 0u};
        bool have_first = false;
        int iso_first = 0;
        if (PROBE) {
            if (lane < P) pk0 = masses[lane];
            if (lane + WAVE < P) pk1 = masses[lane + WAVE];
            if (lane + 2 * WAVE < P) pk2 = masses[lane + 2 * WAVE];
            if (si.z0 <= si.z1) {  // (the same expressions as in the query loops below)
                const bool fold0 = av.sc().min_isotope_err != av.sc().max_isotope_err;
                iso_first = fold0 ? av.sc().min_isotope_err : 0;
                const float precursor_mass = si.mzp * (float)si.z0;
                const Tol ptol = av.sc().wide_window ? tol_scaled(si.iso_tol, (float)si.z0) : SAGE_LOAD_TOL(av.sc().precursor_tol);
                const float center = precursor_mass - (float)iso_first * NEUTRON;  // scoring.rs:344
                q_first = query_window<true>(av.db().pep_mono, av.db().np, ptol, center, av.db().pep_lut, av.db().pep_lut_bins, av.db().pep_lut_inv_w);
                have_first = true;
            }
            if (lane < P) L.win_lo[lane] = pk0;
            if (lane + WAVE < P) L.win_lo[lane + WAVE] = pk1;
            if (lane + 2 * WAVE < P) L.win_lo[lane + 2 * WAVE] = pk2;
        }
        // fragment-tolerance window of every (peak, fragment charge): database.rs:481 on the
        // experimental mass peak*charge of scoring.rs:360
        for (uint32_t i = PROBE ? lane + 3 * WAVE : lane; i < P; i += WAVE) {
            const float m = masses[i];
            if (PROBE) {
                L.win_lo[i] = m;
                continue;
            }
            for (uint32_t fz = 1; fz <= nfz_max; fz++) {
                float lo, hi;
                tol_bounds(ftol0, m * (float)fz, lo, hi);
                L.win_lo[(size_t)(fz - 1) * pcap + i] = lo;
                L.win_hi[(size_t)(fz - 1) * pcap + i] = hi;
            }
        }
        __syncthreads();
        if (pc.slot && lane == 0) pc.bytes(DBG_NARROW_CELLS, 4ull * P);  // (the peak masses)
        bool sorted_ok = true;  // (stream) the sorted-count shortcut needs ascending bounds and lo <= hi
        uint32_t ptop = 0;
        if (!PROBE) {
            bool mono_ok = true;
            for (uint32_t fz = 0; fz < nfz_max; fz++) {
                const float* wl = L.win_lo + (size_t)fz * pcap;
                const float* wh = L.win_hi + (size_t)fz * pcap;
                for (uint32_t i = lane; i < P; i += WAVE) {
                    mono_ok = mono_ok && (wl[i] <= wh[i]);
                    if (i > 0) mono_ok = mono_ok && (wl[i - 1] <= wl[i]) && (wh[i - 1] <= wh[i]);
                }
            }
            sorted_ok = __ballot(!mono_ok) == 0ull;
            ptop = pow2_floor(P);
        }
        pc.mark(0);

        av.refresh();
        const bool fold = av.sc().min_isotope_err != av.sc().max_isotope_err;  // scoring.rs:391
        const int isoA = fold ? av.sc().min_isotope_err : 0, isoB = fold ? av.sc().max_isotope_err : 0;

        UList A, B;  // wave-uniform state
        A.items = L.listA; A.cap = fold ? av.sc().list_cap : 0; A.stored = 0; A.len = 0; A.ok = true;
        B.items = L.listB; B.cap = av.sc().list_cap; B.stored = 0; B.len = 0; B.ok = true;
        bool deferred = false;                     // uniform
        uint32_t tot_matched = 0, tot_scored = 0;  // uniform

        for (uint32_t z = si.z0; z <= si.z1 && !deferred; z++) {
            av.refresh();
            const uint32_t nfz = max_fragment_charge(av.sc().max_fragment_charge, z) - 1;
            const float precursor_mass = si.mzp * (float)z;
            const Tol ptol = av.sc().wide_window ? tol_scaled(si.iso_tol, (float)z) : SAGE_LOAD_TOL(av.sc().precursor_tol);
            if (fold) { A.stored = 0; A.len = 0; }
            for (int iso = isoA; iso <= isoB && !deferred; iso++) {
                const float center = precursor_mass - (float)iso * NEUTRON;  // scoring.rs:344
                av.refresh();
                const Window q = have_first && z == si.z0 && iso == iso_first
                                     ? q_first  // (searched ahead of the staging, above)
                                     : query_window<true>(av.db().pep_mono, av.db().np, ptol, center, av.db().pep_lut, av.db().pep_lut_bins, av.db().pep_lut_inv_w);
                const uint32_t left = q.left;
                const uint32_t potential = q.right - q.left + 1;  // scoring.rs:351
                if (potential > av.sc().wcap) {
                    deferred = true;
                    break;
                }
                pc.mark(1);
                if (pc.slot && lane == 0) { atomicAdd(&pc.slot[6], (unsigned long long)potential); atomicAdd(&pc.slot[7], 1ull); }
                cnt.zero(potential, lane);
                __syncthreads();
                uint32_t acc = 0;
                if (!PROBE && q.first < q.end) {
                    av.refresh();
                    const uint64_t f0 = av.db().pm_off[q.first], f1 = av.db().pm_off[q.end];
                    const SageTheoretical* __restrict__ pm_frag = av.db().pm_frag;
                    const uint32_t pcap = av.b().pcap;
                    auto count_one = [&](float frag) -> uint32_t {
                        if (!sorted_ok) {
                            uint32_t c = 0;
                            for (uint32_t fz = 0; fz < nfz; fz++)
                                c += count_windows_scan(L.win_lo + (size_t)fz * pcap, L.win_hi + (size_t)fz * pcap, P, frag);
                            return c;
                        }
                        switch (nfz) {
                            case 1: return count_windows_lockstep<1>(L.win_lo, L.win_hi, pcap, P, ptop, frag);
                            case 2: return count_windows_lockstep<2>(L.win_lo, L.win_hi, pcap, P, ptop, frag);
                            case 3: return count_windows_lockstep<3>(L.win_lo, L.win_hi, pcap, P, ptop, frag);
                            default: {
                                uint32_t c = 0;
                                for (uint32_t fz = 0; fz < nfz; fz++)
                                    c += count_windows_sorted(L.win_lo + (size_t)fz * pcap, L.win_hi + (size_t)fz * pcap, P, frag);
                                return c;
                            }
                        }
                    };
                    // software-pipelined stream over the window's fragments: the next trip's two 8-byte
                    // loads are issued before this trip's LDS searches
                    SageTheoretical n0{0, 0.f}, n1{0, 0.f};
                    uint64_t j = f0 + lane;
                    if (j < f1) n0 = pm_frag[j];
                    if (j + WAVE < f1) n1 = pm_frag[j + WAVE];
                    for (; j < f1; j += 2 * WAVE) {
                        const SageTheoretical fr0 = n0, fr1 = n1;
                        const bool has1 = j + WAVE < f1;
                        const uint64_t jn = j + 2 * WAVE;
                        if (jn < f1) n0 = pm_frag[jn];
                        if (jn + WAVE < f1) n1 = pm_frag[jn + WAVE];
                        const uint32_t c0 = count_one(fr0.fragment_mz);
                        const uint32_t c1 = has1 ? count_one(fr1.fragment_mz) : 0;
                        if (c0) { cnt.add(fr0.peptide_index - left, c0); acc += c0; }
                        if (c1) { cnt.add(fr1.peptide_index - left, c1); acc += c1; }
                    }
                }
                if (PROBE && q.first < q.end) {
                    // Per (peak, fragment charge) window, two reads of the (small) tile's position table give the RUN of index
                    // entries inside the window's table cells; the entries of the run are tested against the fragment and
                    // precursor windows (database.rs:526-533) and hits bump the LDS counters.  Run lengths are very skewed
                    // (fragment masses sit in narrow mass-defect bands: most windows of a spectrum find nothing, a few find
                    // dozens of entries), so the runs of a batch of windows are FLATTENED into a list of 16-byte cells (two
                    // entries each) shared evenly by the 64 lanes, each lane's cells in flight together: one round trip for
                    // the table, one for the entries, no lane waiting on another lane's long run.
                    av.refresh();
                    const uint4* __restrict__ frag2 = (const uint4*)av.db().tm2_frag;
                    const LutWord* __restrict__ tm2_l1 = av.db().tm2_l1;
                    const uint32_t* __restrict__ tm2_pos = av.db().tm2_pos;
                    const uint32_t tile2_shift = av.db().tile2_shift, lut2_stride = av.db().lut2_stride, lut2_words = av.db().lut2_words;
                    const float lut2_scale = av.db().lut2_scale;
                    const Tol ftol = SAGE_LOAD_TOL(av.sc().fragment_tol);
                    const uint32_t t0 = q.first >> tile2_shift, t1 = (q.end - 1) >> tile2_shift;
                    const uint32_t nprobe = P * nfz;
                    uint32_t* const tp0 = L.ptab;                      // [PROBE_BATCH] first entry of the run
                    uint32_t* const tp1 = L.ptab + PROBE_BATCH;        // [PROBE_BATCH] end entry
                    uint32_t* const tcs = L.ptab + 4 * PROBE_BATCH;    // [PROBE_TCS_WORDS] first cell in the flattened list, then sentinels
                    // [PROBE_BATCH] each: the window's bounds, worked out once where its table cells are (round 6: a cell's lane
                    // repeated window_of — an integer division by P, Tolerance::bounds' two divisions — for every cell it walked)
                    float* const tlo = (float*)(L.ptab + 2 * PROBE_BATCH);
                    float* const thi = (float*)(L.ptab + 3 * PROBE_BATCH);
                    if (PROBE_TCS_WORDS > PROBE_BATCH && lane < PROBE_TCS_WORDS - PROBE_BATCH) tcs[PROBE_BATCH + lane] = NO_WINDOW;  // (never rewritten)
#ifndef SAGE_PROBE_SYM
#define SAGE_PROBE_SYM 1      // a symmetric ppm tolerance: one division per window (core.h: tol_bounds_sym, the same bits)
#endif
#ifndef SAGE_PROBE_FASTDIV
#define SAGE_PROBE_FASTDIV 1  // ... and that division in the short form where every lane's dividend is in its proven range
#endif
                    const bool sym_ppm = SAGE_PROBE_SYM && (av.sc().tol_mode & TOL_SYM) != 0u && ftol.kind == 0;
                    auto window_of = [&](uint32_t pr, float& flo, float& fhi) {
                        flo = 1.0f; fhi = 0.0f;  // (no such window: empty)
                        const bool has = pr < nprobe;
                        float center = 0.f;
                        if (has) {
                            const uint32_t fz = pr / P, i = pr - fz * P;
                            center = L.win_lo[i] * (float)(fz + 1);
                        }
                        if (sym_ppm) {  // (wave-uniform)
                            // x / 1e6 of core.h: div_1e6_fast is the IEEE quotient for every x of FAST_DIV_LO <= |x| <= FAST_DIV_HI
                            // (tests/hostemu/div_const_proof.c walks all 2^32 dividends); here the dividends are the caller's peaks,
                            // which no host code has bounded — so the lanes look: one window outside (a zero, a NaN, an infinity)
                            // and the whole wavefront takes the IEEE sequence for this batch
                            const float x = center * ftol.hi, ax = __builtin_fabsf(x);
                            float d;
                            if (SAGE_PROBE_FASTDIV && __ballot(has && !(ax >= FAST_DIV_LO && ax <= FAST_DIV_HI)) == 0ull) d = div_1e6_fast(x);
                            else d = x / 1000000.0f;
                            if (has) { flo = center + -d; fhi = center + d; }
                        } else if (has) {
                            tol_bounds(ftol, center, flo, fhi);
                        }
                    };
                    for (uint32_t t = t0; t <= t1; t++) {  // (one tile unless the window straddles a tile boundary)
                        const LutWord* __restrict__ l1 = tm2_l1 + (size_t)t * lut2_words;
                        // table reads of up to PROBE_BATCH windows, PROBE_PER_LANE per lane, all in flight together; window q of the
                        // flattened order (lane-major) is window pbase + (q % PPL) * 64 + q / PPL.  The reads of batch b + 1 are
                        // issued before the cells of batch b are walked (SAGE_PROBE_PIPELINE): one round trip less per further batch.
#ifndef SAGE_PROBE_PIPELINE
#define SAGE_PROBE_PIPELINE 1
#endif
                        uint32_t np0[PROBE_PER_LANE], np1[PROBE_PER_LANE];
                        float nlo[PROBE_PER_LANE], nhi[PROBE_PER_LANE];
                        // The table in succinct form (core.h: LutWord): a window's run is [pos[rank(icl)], pos[rank(ich)]), the ranks from
                        // the occupancy word(s) of its cells — 40 KB per tile, resident in the caches while the tile's spectra are
                        // scored — and equal ranks say "empty" (most windows) without a second read.  The words of all of a lane's
                        // windows first, then the run starts of the non-empty ones: two dependent trips, the first a short one.
                        auto issue = [&](uint32_t pbase) {
                            uint32_t icl[PROBE_PER_LANE], ich[PROBE_PER_LANE];
                            LutWord wa[PROBE_PER_LANE], wb[PROBE_PER_LANE];
#pragma unroll
                            for (uint32_t i = 0; i < PROBE_PER_LANE; i++) {
                                window_of(pbase + i * WAVE + lane, nlo[i], nhi[i]);
                                // (core.h: the scale is a power of two, no safety margin needed)
                                lut_cells(nlo[i], nhi[i], lut2_scale, lut2_stride, icl[i], ich[i]);
                                wa[i] = l1[icl[i] >> 5];
                                wb[i] = l1[ich[i] >> 5];
                            }
#pragma unroll
                            for (uint32_t i = 0; i < PROBE_PER_LANE; i++) {
                                const uint32_t r0 = lut_rank(wa[i], icl[i]), r1 = lut_rank(wb[i], ich[i]);
                                np0[i] = np1[i] = 0;
                                if (r1 > r0) {
                                    np0[i] = tm2_pos[r0];
                                    np1[i] = tm2_pos[r1];
                                }
                            }
                        };
                        if (SAGE_PROBE_PIPELINE) issue(0);
                        for (uint32_t pbase = 0; pbase < nprobe; pbase += PROBE_BATCH) {
                            if (!SAGE_PROBE_PIPELINE) issue(pbase);
                            uint32_t rp0[PROBE_PER_LANE], rp1[PROBE_PER_LANE];
#pragma unroll
                            for (uint32_t i = 0; i < PROBE_PER_LANE; i++) {
                                rp0[i] = np0[i]; rp1[i] = np1[i];
                                tlo[lane * PROBE_PER_LANE + i] = nlo[i];  // (the previous batch's cells are done: wave_sync below)
                                thi[lane * PROBE_PER_LANE + i] = nhi[i];
                            }
                            if (SAGE_PROBE_PIPELINE && pbase + PROBE_BATCH < nprobe) issue(pbase + PROBE_BATCH);
                            uint32_t tot = 0;
#pragma unroll
                            for (uint32_t i = 0; i < PROBE_PER_LANE; i++) tot += rp1[i] > rp0[i] ? ((rp1[i] - 1) >> 1) - (rp0[i] >> 1) + 1 : 0;
                            const uint32_t incl = wave_incl_scan_dpp(tot);  // inclusive prefix of the lanes' cell counts
                            uint32_t run = incl - tot;
#pragma unroll
                            for (uint32_t i = 0; i < PROBE_PER_LANE; i++) {
                                tp0[lane * PROBE_PER_LANE + i] = rp0[i];
                                tp1[lane * PROBE_PER_LANE + i] = rp1[i];
                                tcs[lane * PROBE_PER_LANE + i] = run;
                                run += rp1[i] > rp0[i] ? ((rp1[i] - 1) >> 1) - (rp0[i] >> 1) + 1 : 0;
                            }
                            const uint32_t n_cells = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                            if (pc.slot) {  // (profiling instance: bytes asked for — two 8-byte occupancy words per window, two run starts per non-empty one)
                                uint32_t nonempty = 0;
#pragma unroll
                                for (uint32_t i = 0; i < PROBE_PER_LANE; i++) nonempty += (uint32_t)__popcll(__ballot(rp1[i] > rp0[i]));
                                if (lane == 0) {
                                    const uint32_t nw = nprobe - pbase < PROBE_BATCH ? nprobe - pbase : PROBE_BATCH;
                                    pc.bytes(DBG_NARROW_LUT, 16ull * nw + 8ull * nonempty);
                                    pc.bytes(DBG_NARROW_CELLS, 16ull * n_cells);
                                }
                            }
                            wave_sync();
                            for (uint32_t cb = 0; cb < n_cells; cb += WAVE * PROBE_CELLS) {
                                uint4 e[PROBE_CELLS];
                                uint32_t eq[PROBE_CELLS], ej[PROBE_CELLS];
#pragma unroll
                                for (uint32_t c = 0; c < PROBE_CELLS; c++) {
                                    const uint32_t k = cb + c * WAVE + lane;
                                    e[c] = make_uint4(0u, 0u, 0u, 0u);
                                    eq[c] = NO_WINDOW;
                                    ej[c] = 0;
                                    if (k < n_cells) {
                                        uint32_t a = 0;  // the owner of cell k: the largest q with tcs[q] <= k (tcs ascends; windows
                                                         // with an empty run share their successor's start)
#pragma unroll
                                        for (uint32_t step = PROBE_TCS_WORDS / 2; step; step >>= 1) a += (tcs[a + step] <= k) ? step : 0;
                                        eq[c] = a;
                                        ej[c] = ((tp0[a] >> 1) + (k - tcs[a])) << 1;
                                        e[c] = frag2[ej[c] >> 1];  // (tm2_frag is padded by 2 entries)
                                    }
                                }
#pragma unroll
                                for (uint32_t c = 0; c < PROBE_CELLS; c++) {
                                    if (eq[c] == NO_WINDOW) continue;
                                    const uint32_t a = eq[c];
                                    const float lo = tlo[a], hi = thi[a];
                                    const uint32_t p0 = tp0[a], p1 = tp1[a], j = ej[c];
                                    const float mz0 = __uint_as_float(e[c].y), mz1 = __uint_as_float(e[c].w);
                                    if (j >= p0 && j < p1 && mz0 >= lo && mz0 <= hi && e[c].x >= q.first && e[c].x < q.end) { cnt.add(e[c].x - left, 1); acc++; }
                                    if (j + 1 < p1 && mz1 >= lo && mz1 <= hi && e[c].z >= q.first && e[c].z < q.end) { cnt.add(e[c].z - left, 1); acc++; }
                                }
                            }
                            wave_sync();  // (the next batch rewrites the run table)
                        }
                    }
                }
                const uint32_t matched = wave_sum(acc);
                __syncthreads();
                pc.mark(2);
                av.refresh();
                const uint32_t kmax = av.sc().kmax, report_psms = av.sc().report_psms;
                res.q_left = left;            // (of the last query: what prelim_kernel keeps of a single-query spectrum)
                res.q_potential = potential;
                tot_matched += matched;
                UList& target = fold ? A : B;
                if (matched == 0) {  // scoring.rs:376-378: the untrimmed all-default vector
                    ulist_append_empties(target, potential, kmax);
                    continue;
                }
                // ---- trim_hits of this query, scoring.rs:380 ----
                const uint32_t k = trim_k(potential, report_psms);
                uint32_t scored = 0;
                if (potential <= k) {  // no k-select: the slots go to the list verbatim
                    for (uint32_t base = 0; base < potential; base += WAVE) {
                        const uint32_t i = base + lane;
                        const uint32_t c = i < potential ? cnt.get(i) : 0;
                        scored += (uint32_t)__popcll(__ballot(c > 0));
                        const uint32_t nvalid = potential - base < WAVE ? potential - base : WAVE;
                        ulist_append(target, c ? pack_prescore(c, left + i, z, iso) : PRESCORE_EMPTY, nvalid, kmax);
                    }
                } else if (BIGK) {
                    // k > 64 (report_psms > 32): the heap in LDS (always exact; see lh_build)
                    res.untrimmed = false;
                    uint64_t* hp = L.heap;
                    for (uint32_t base = 0; base < k; base += WAVE) {
                        const uint32_t i = base + lane;
                        const uint32_t c = i < k ? cnt.get(i) : 0;
                        if (i < k) hp[i] = c ? pack_prescore(c, left + i, z, iso) : PRESCORE_EMPTY;
                        scored += (uint32_t)__popcll(__ballot(c > 0));
                    }
                    lh_build(hp, k);
                    for (uint32_t base = k; base < potential; base += WAVE) {
                        const uint32_t i = base + lane;
                        const uint32_t c = i < potential ? cnt.get(i) : 0;
                        scored += (uint32_t)__popcll(__ballot(c > 0));
                        lh_offer_batch(hp, k, pack_prescore(c, left + i, z, iso), c > 0);
                    }
                    for (uint32_t base = 0; base < k; base += WAVE)
                        ulist_append(target, base + lane < k ? hp[base + lane] : PRESCORE_EMPTY, k - base < WAVE ? k - base : WAVE, kmax);
                } else if (!exact && fast_select(L, cnt, potential, k, left, z, iso, kmax, target, scored)) {
                    // (done: the k largest slots by (count, slot) without replaying the heap)
                    res.untrimmed = false;
                } else {
                    // keys `matched << 16 | slot` (potential <= wcap <= 65536): same order as PreScore inside one query
                    res.untrimmed = false;
                    scored = 0;
                    Heap32 hp;
                    {
                        const uint32_t c = lane < k ? cnt.get(lane) : 0;
                        wh32_init(hp, c ? (c << 16) | lane : 0u, k);
                        scored += (uint32_t)__popcll(__ballot(c > 0));
                    }
                    wh32_build(hp, k);
                    for (uint32_t base = k; base < potential; base += WAVE) {
                        const uint32_t i = base + lane;
                        const uint32_t c = i < potential ? cnt.get(i) : 0;
                        const uint32_t v = (c << 16) | i;
                        // in slot order; a slot can only enter if its count reaches the heap minimum's (heap.rs:22: later
                        // slots have larger peptide indices, so equal counts do enter)
                        uint64_t mask = __ballot(c > 0 && c >= (wh32_get(hp, 0) >> 16));
                        scored += (uint32_t)__popcll(__ballot(c > 0));
                        if (pc.slot && lane == 0) atomicAdd(&pc.slot[5], (unsigned long long)__popcll(mask));
                        while (mask) {
                            const uint32_t bit = (uint32_t)__ffsll((long long)mask) - 1;
                            mask &= mask - 1;
                            wh32_offer(hp, k, (uint32_t)__builtin_amdgcn_readlane((int)v, (int)__builtin_amdgcn_readfirstlane(bit)));
                        }
                    }
                    const uint32_t h = hp.h;
                    ulist_append(target, h ? pack_prescore(h >> 16, left + (h & 0xFFFFu), z, iso) : PRESCORE_EMPTY, k, kmax);
                }
                tot_scored += scored;
                __syncthreads();
                pc.mark(3);
            }
            if (fold && !deferred) {  // scoring.rs:405 then `hits +=` at :432 / :450
                av.refresh();
                const uint32_t kmax = av.sc().kmax, report_psms = av.sc().report_psms;
                if (A.len > trim_k(A.len, report_psms)) res.untrimmed = false;
                if (BIGK) ulist_trim_big(A, report_psms);
                else ulist_trim(A, report_psms, exact || av.sc().list_cap > 4 * WAVE);
                __syncthreads();
                for (uint32_t base = 0; base < A.stored; base += WAVE) {
                    const uint64_t v = base + lane < A.stored ? A.items[base + lane] : PRESCORE_EMPTY;
                    const uint32_t nvalid = A.stored - base < WAVE ? A.stored - base : WAVE;
                    ulist_append(B, v, nvalid, kmax);
                }
                __syncthreads();
            }
        }
        if (deferred) {  // some precursor window of this spectrum is too large for the LDS counters
            res.deferred = true;
            return res;
        }
        av.refresh();
        const uint32_t report_psms = av.sc().report_psms;
        if (B.len > trim_k(B.len, report_psms)) res.untrimmed = false;
        if (BIGK) ulist_trim_big(B, report_psms);
        else ulist_trim(B, report_psms, exact || av.sc().list_cap > 4 * WAVE);  // scoring.rs:460
        __syncthreads();
        res.stored = B.stored;
        res.matched = tot_matched;
        res.scored = tot_scored;
        res.ok = A.ok && B.ok;
    }
    return res;
}

// HUGE (wide lists only): the lists and the heap in the workgroup's slice of DevWork::hugebuf — an instance of its own, so that the
// LDS instance keeps LDS pointers (one instance choosing at run time makes them generic: flat accesses, 3x the kernel time)
template <bool PROBE, bool PROF, bool BIGK = false, bool HUGE = false>
__global__ __launch_bounds__(64) SAGE_PRELIM_WAVES_ATTR void prelim_kernel(PrelimKernargs A) {
    // (ONE argument: prelim_spectrum and the output below read what they need from the kernarg segment, phase by phase — ArgRef)
    const DevDbView& db = A.db;
    const DevScorer& sc = A.sc;
    const DevBatchView& b = A.b;
    const DevWork& w = A.w;
    typedef typename std::conditional<PROF, PhaseClock, NoClock>::type Clock;
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t lane = lane_id();
    staggered_start(blockIdx.x);
    const PrelimLds L = carve_prelim(smem, sc, b, BIGK && HUGE ? w.hugebuf + (size_t)blockIdx.x * w.huge_stride : nullptr);

    uint32_t n_batch = b.n;
    if (b.n_dev) {  // retry pass: the count is a device-side counter of the first pass
        const uint32_t nd = uni(*b.n_dev);
        n_batch = nd < n_batch ? nd : n_batch;
    }
    for (uint32_t blk = blockIdx.x; blk < n_batch; blk += gridDim.x) {
        const uint32_t pos = xcd_position(blk, n_batch, sc.xcd_chunk);
        uint32_t spec = b.sched ? 0u : b.order ? uni(b.order[pos]) : pos;
        __syncthreads();
        Clock pc;
        pc.start((sc.dbg_flags & 512u) && !sc.exact ? nullptr : w.dbg, blk, 0);  // (SAGE_HIP_DEBUG_FLAGS=512: clocks of the exact retry pass only)
        const SpecInfo si = b.sched ? load_spec_sched(sc, b, pos, spec) : load_spec(sc, b, spec);
        const PrelimResult r = prelim_spectrum<PROBE, BIGK, PrelimKernargs>(db, sc, b, L, si, sc.exact != 0, pc);
        // ---- the results leave: the work-set pointers come from the kernarg segment HERE (they were not kept across the matching) ----
        typedef const __attribute__((address_space(4))) PrelimKernargs* Segment;
        Segment ka = (Segment)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        uint32_t* const w_cnt_store = ka->w.cnt_store;
        if (!BIGK && w_cnt_store) {
            // One precursor-window query (known charge, one isotope error): its window counts — still in LDS — stay in HBM for
            // rescore_kernel's tie settlement.  (Here, behind prelim_spectrum, not inside it: nothing of this is live in the matching loops.)
            // Rows in SCHEDULE order (row `pos`, not row `spec`): the wavefronts running at any time write one moving window of a
            // few MB, not 2 KB pieces scattered over the whole GB — the stores of a random row per wavefront cost 5 % of the kernel
            // (address translation).  A row: {left, potential, -, -} then the u16 counts, two per word.
            const bool keep = !r.deferred && si.z0 == si.z1 && ka->sc.min_isotope_err == ka->sc.max_isotope_err;
            uint32_t* __restrict__ row = w_cnt_store + (size_t)pos * ka->w.cnt_stride;
#ifndef SAGE_CNT_MODE
#define SAGE_CNT_MODE 0  // (measurement variants: 1 = the row header only, 2 = the counts only)
#endif
            if (keep && SAGE_CNT_MODE != 1)
                for (uint32_t i = lane; i < (r.q_potential + 1) / 2; i += WAVE) row[CNT_ROW_HEADER + i] = L.cnt[i];
            if (lane == 0 && SAGE_CNT_MODE != 2) {
                row[0] = keep ? r.q_left : 0u;
                row[1] = keep ? r.q_potential : 0u;  // (0: no counts kept — several queries, or a large window)
            }
        }
        if (r.deferred) {
            if (lane == 0) {
                ka->w.status[spec] = ST_DEFERRED;
                // (queue_later: queue_kernel appends the marked spectra behind this kernel.  One returning atomic per spectrum on ONE
                // address is ~11 ns each whatever else the workgroup does: 2.3 ms of C5's step, where every spectrum ends up here)
                if (!ka->w.queue_later) {
                    const uint32_t it = atomicAdd(ka->w.n_deferred + CTR_QUEUED, 1u);
                    ka->w.queue[it] = spec;
                    if (!ka->w.reuse) ka->w.item_of[spec] = it;  // (the retry pass finds the first pass's records of this spectrum through it)
                }
            }
            continue;
        }
        if (lane == 0) {
            if (!r.ok) atomicAdd(ka->w.n_deferred + CTR_LIST_OVERFLOW, 1u);
            ka->w.status[spec] = r.ok ? (r.untrimmed ? ST_OK_ORDERED : ST_OK) : ST_OVERFLOW;
            ka->w.cand_len[spec] = r.stored;
            ka->w.totals[2 * spec] = r.matched;
            ka->w.totals[2 * spec + 1] = r.scored;
        }
        {
            uint64_t* __restrict__ dst = ka->w.cand + (size_t)spec * ka->sc.kmax;
            for (uint32_t i = lane; i < r.stored; i += WAVE) dst[i] = L.listB[i];
        }
        pc.mark(4);
    }
}

// The queue of the large-window kernels behind a prelim_kernel that only marked the spectra it hands over (DevWork::queue_later): one
// thread per schedule position, a wavefront's marked spectra per atomic on the queue's counter.
__global__ __launch_bounds__(256) void queue_kernel(DevScorer sc, DevBatchView b, DevWork w) {
    uint32_t n_batch = b.n;
    if (b.n_dev) {
        const uint32_t nd = *b.n_dev;
        n_batch = nd < n_batch ? nd : n_batch;
    }
    const uint32_t lane = lane_id();
    for (uint32_t base = blockIdx.x * blockDim.x; base < n_batch; base += gridDim.x * blockDim.x) {  // (whole wavefronts stay together)
        const uint32_t blk = base + threadIdx.x;
        uint32_t spec = 0;
        bool marked = false;
        if (blk < n_batch) {
            // (ascending precursor mass, whatever prelim_kernel's own schedule: neighbours in the queue share their tiles —
            // C5 43.8 ms in the schedule's XCD-chunked order, 43.2 ms so)
            const uint32_t pos = blk;
            spec = b.order ? b.order[pos] : pos;
            marked = w.status[spec] == ST_DEFERRED;
        }
        const uint64_t m = __ballot(marked);
        if (m == 0ull) continue;
        uint32_t first = 0;
        if (lane == 0) first = atomicAdd(w.n_deferred + CTR_QUEUED, (uint32_t)__popcll(m));
        first = uni(first);
        if (marked) {
            const uint32_t it = first + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            w.queue[it] = spec;
            if (!w.reuse) w.item_of[spec] = it;
        }
    }
}

// ---- large windows (open search, wide-window / DIA, mid-size tolerances) ---------------------------
// A window of 10^3..10^6 candidates is walked in TILES of 2^tile_shift consecutive peptides.  The index has a
// tile-major copy (device_types.h): inside a tile the fragments are m/z-sorted and a position table turns a
// fragment-tolerance window into a short contiguous run.  The kernels:
//   count    : persistent workgroups pull queued spectra.  Per tile, every (peak, fragment charge) window reads its two
//              table words, the runs are cut into 16-byte cells (two entries) that all 512 threads share evenly, and each
//              entry inside both the m/z window and the precursor window bumps a counter of the tile IN LDS (u8 in the
//              first pass of a search, u16 otherwise; no counter traffic to HBM) with one returning atomic whose old value
//              also maintains the histogram of counts and marks the slot in a bitmap once it reaches the tile's pruning
//              threshold.  After each tile the wavefronts append the marked slots, in slot order, to the query's candidate
//              directory in HBM.  "Pruning threshold": the k-th largest count of all EARLIER tiles, a lower bound of the
//              heap minimum at that point of heap.rs:21-27 — pruning with it never changes the replay.
//   select   : order-free trim_hits: the k candidates bounded_min_heapify would keep, without the heap.
//   replay   : trim_hits' bounded_min_heapify (heap.rs:7-28) itself, sequential per query: one wavefront per query (heap one
//              element per lane) or one lane per query (heap in LDS, 64 queries per wavefront).
//   assemble : one wavefront per spectrum concatenates / folds the per-query lists exactly as
//              scoring.rs:384-462 does and writes the final preliminary list.
// Same predicate as database.rs:526-533, so counts — and everything downstream — are identical.
constexpr uint32_t TILE_THREADS = 512;
constexpr uint32_t TILE_WAVES = TILE_THREADS / WAVE;
constexpr uint32_t HIST_BINS = 64;
constexpr uint32_t NONE32 = 0xFFFFFFFFu;

struct TileLds {
    uint32_t* cnt;      // [tile_size / 2] u16 pairs, or [tile_size / 4] u8 quads (cnt8)
    uint32_t* bm;       // [tile_size / 32] one bit per slot: its count reached this tile's pruning threshold (a candidate)
    float* win_lo;      // [fzcap * pcap]
    float* win_hi;
    uint32_t* hist;     // [HIST_BINS] non-empty slots of the query so far, by matched count
    uint32_t* sh;       // [16] workgroup-shared scalars
    // the runs of the unit being streamed (one window per thread): first entry, end entry, first 16-byte cell of the run in the
    // wavefront's flattened cell list; psum[w] = cells of wavefront w's windows
    uint32_t* pp0;      // [TILE_THREADS]
    uint32_t* pp1;      // [TILE_THREADS]
    uint32_t* pcs;      // [TILE_THREADS] first cell of the window's run in the UNIT's flattened cell list
    uint32_t* psum;     // [TILE_WAVES] cells of wavefront w's windows
    uint32_t* nz;       // [TILE_WAVES] windows of wavefront w with a non-empty run
    // where the runs start (round 6: a cell's owner by rank, not by search — tile_count_body: locate): per block of 64 flattened cells
    // one 64-bit word, bit j = a run starts at cell 64 b + j (two sets, units alternate: the idle one is cleared while the other is
    // marked), and the rank — index among the unit's non-empty runs — of the run that owns the block's first cell; cpid: rank -> window
    uint32_t* mbits;    // [2][mark_blocks][2]
    uint32_t* mbase;    // [mark_blocks]
    uint16_t* cpid;     // [TILE_THREADS]
    uint32_t* qw;       // [TILE_QW_MAX * 4] {left, right, first, end} of the spectrum's precursor-window queries, searched up front
};
constexpr uint32_t TILE_QW_MAX = 64;  // queries per spectrum whose windows are searched up front, a wavefront each (more: one by one)
// blocks of 64 flattened cells the run-start marks of a unit cover at a time (a unit with more cells re-marks, a rare and slow path):
// the u8 instance keeps 3 cells per thread in flight (1 536 per round, 24 blocks), the u16 one 4 (2 048, 32 blocks) and has the LDS of
// two workgroups per compute unit to fit
#ifndef SAGE_MARK_BLOCKS8
#define SAGE_MARK_BLOCKS8 96
#endif
#ifndef SAGE_MARK_BLOCKS16
#define SAGE_MARK_BLOCKS16 40
#endif
__host__ __device__ constexpr uint32_t tile_mark_blocks(bool cnt8) { return cnt8 ? SAGE_MARK_BLOCKS8 : SAGE_MARK_BLOCKS16; }
// `wing`: the windows of the spectrum live in a global-memory workspace instead (tile_count_wing_kernel: spectra whose peaks x
// fragment charges do not fit a compute unit's LDS next to the counters)
__host__ __device__ inline size_t tile_lds_layout(uint32_t tile_shift, const DevBatchView& b, TileLds* l, unsigned char* smem, bool cnt8,
                                                  bool wing = false) {
    size_t off = 0;
    if (l) l->cnt = (uint32_t*)(smem + off);
    off += ((size_t)1 << tile_shift) * (cnt8 ? 1 : 2);
    if (l) l->bm = (uint32_t*)(smem + off);
    off += (((size_t)1 << tile_shift) / 32 + 3) / 4 * 16;
    if (l) l->win_lo = (float*)(smem + off);
    off += wing ? 0 : (size_t)b.fzcap * b.pcap * 4;
    if (l) l->win_hi = (float*)(smem + off);
    off += wing ? 0 : (size_t)b.fzcap * b.pcap * 4;
    off = (off + 15) & ~(size_t)15;  // (what follows is read in 16-byte pieces)
    if (l) l->hist = (uint32_t*)(smem + off);
    off += HIST_BINS * 4;
    if (l) l->sh = (uint32_t*)(smem + off);
    off += 16 * 4;
    if (l) l->pp0 = (uint32_t*)(smem + off);
    off += TILE_THREADS * 4;
    if (l) l->pp1 = (uint32_t*)(smem + off);
    off += TILE_THREADS * 4;
    if (l) l->pcs = (uint32_t*)(smem + off);
    off += TILE_THREADS * 4;
    if (l) l->psum = (uint32_t*)(smem + off);
    off += TILE_WAVES * 4;
    if (l) l->nz = (uint32_t*)(smem + off);
    off += TILE_WAVES * 4;
    static_assert((2 * TILE_WAVES * 4) % 16 == 0, "psum, nz, mbits, mbase, cpid back to back (tile_count_body derives them from psum)");
    if (l) l->mbits = (uint32_t*)(smem + off);
    off += (size_t)2 * tile_mark_blocks(cnt8) * 8;
    if (l) l->mbase = (uint32_t*)(smem + off);
    off += (size_t)tile_mark_blocks(cnt8) * 4;
    if (l) l->cpid = (uint16_t*)(smem + off);
    off += TILE_THREADS * 2;
    off = (off + 15) & ~(size_t)15;
    if (l) l->qw = (uint32_t*)(smem + off);
    off += 64 * 4 * 4;  // (TILE_QW_MAX, declared below the struct)
    return (off + 15) & ~(size_t)15;
}

enum { SH_ITEM = 0, SH_LEFT, SH_RIGHT, SH_FIRST, SH_END, SH_MATCHED, SH_SCORED, SH_DIR, SH_ARENA_OK, SH_CHUNK_CUR, SH_CHUNK_LIM, SH_THR, SH_OVF, SH_NCAND };
constexpr uint32_t ARENA_CHUNK = 1u << 16;  // entries a workgroup takes from the global arena at a time

__device__ __forceinline__ uint32_t query_index(const DevScorer& sc, const SpecInfo& si, uint32_t z, int iso) {
    const bool fold = sc.min_isotope_err != sc.max_isotope_err;
    const uint32_t n_iso = fold ? (uint32_t)(sc.max_isotope_err - sc.min_isotope_err) + 1 : 1;
    return (z - si.z0) * n_iso + (uint32_t)(iso - (fold ? sc.min_isotope_err : 0));
}

constexpr uint32_t CELLS_PER_THREAD = 4;  // 16-byte index cells a thread of the count kernel keeps in flight (x 512 threads per unit)

// The candidate stream of a query is a DIRECTORY in the arena: for every (tile of the window, wavefront of the count kernel)
// one entry {position, count} of a run of candidate words `matched count << 16 | slot in tile`, in slot order inside the run;
// directory order == slot order (tile-major, then the 8 slot ranges of a tile the 8 wavefronts own).  A word with count 0 is
// a hole (a slot that belongs to the verbatim head of the window): readers skip it.
constexpr uint32_t DIR_WORDS = 2;

// Parameters BY VALUE: pointer members of a by-value kernel argument are known to be global, so the compiler emits global_*
// instructions.  (Read through a TileParams pointer they were generic pointers -> FLAT loads / stores / atomics, which count
// on lgkmcnt as well as vmcnt: every `s_waitcnt lgkmcnt(0)` in front of an LDS access then also waited for all outstanding HBM
// loads.)
//
// Two instances.  C8 = false: u16 counters, two per LDS word — 78 KB per workgroup, 2 workgroups per CU.  C8 = true: u8
// counters, four per word — 46 KB, 3 workgroups per CU (6 wavefronts per SIMD for a kernel that is bound by latency).  A u8
// counter that reaches 255 flags its query (SH_OVF): the spectrum is not trusted and goes to the retry pass, which always counts
// in u16 — so the first pass of a search may use u8 and nothing else does (DevWork::cnt8).
template <bool C8, bool WING = false>
__device__ __forceinline__ void tile_count_body(const TileParams& kp, unsigned char* smem) {
    constexpr uint32_t SPW = C8 ? 4 : 2;            // slots per counter word
    constexpr uint32_t CSH = C8 ? 2 : 1;            // slot -> word
    constexpr uint32_t CBITS = C8 ? 8 : 16;         // bits per counter
    constexpr uint32_t CMAX = C8 ? 0xFFu : 0xFFFFu;
#ifndef SAGE_TILE8_CELLS
#define SAGE_TILE8_CELLS 2  // (round 6, after the rank locate: 2 cells per thread in flight spill 7 vector registers instead of 18 — C5 41.8 -> 40.8 ms, C4 unchanged; 3 before)
#endif
// Round 5 experiments on the count kernel's instruction count (it executes vector instructions 80 % of its SIMDs' time), A/B'd on C4
// / C5 (scripts/experiments/r05_lab/gpu_r5q.sh, gpu_r5r.sh): the owner wavefront of a cell by BISECTION over the running totals (14
// instructions instead of 35) and both index ranges of an entry as one unsigned compare each — 20 % SLOWER on C4 (58.0 -> 69.6 ms:
// the selects spill, scratch 128 -> 160 bytes per lane); a linear walk over the running totals (21 instructions) with the old
// predicates: -1 % (57.0 ms), the default now; with the one-compare ranges: 0.  Fewer instructions buy nothing here unless the
// register allocation holds still.
#ifndef SAGE_HIT_RANGES
#define SAGE_HIT_RANGES 1  // (round 6, with the arguments out of the registers: the allocation holds still — 64 / 18 spills either way —, C4 -1 %, C5 -0.7 %)
#endif
#ifndef SAGE_SCAN_SKIP
#define SAGE_SCAN_SKIP 1  // (a wavefront without candidate bits in a tile skips the prefix sum of the scan: C4 -0.4 %, C5 -0.6 %)
#endif
#ifndef SAGE_LOCATE_LINEAR
#define SAGE_LOCATE_LINEAR 1
#endif
    // cells a thread keeps in flight: the u8 instance runs 6 wavefronts per SIMD in 80 VGPRs, the u16 one 4 in 128
    constexpr uint32_t CPT = C8 ? SAGE_TILE8_CELLS : CELLS_PER_THREAD;
    // Arguments where they are used (the ArgRef idea of prelim_kernel): the four views are ~110 scalar registers' worth of pointers
    // and parameters in a kernel that has 100 and keeps a dozen phase-local scalars live per tile.  Read once at the entry they were
    // spilled to VGPR lanes at once and read back tile after tile — v_readlane / scratch loads inside the tile loop, vector-ALU issue
    // slots of a kernel that is bound by exactly those (round 5: 143 scalar + 33 vector spills, 128 bytes of scratch).  A kernel's
    // arguments never need saving: they sit in the kernarg segment, one scalar load away.  `ka` is that segment behind a pointer the
    // compiler cannot see through; TILE_ARGS() re-derives it at the head of every phase, so the loads stay in the phase that uses them.
    typedef const __attribute__((address_space(4))) TileParams* Segment;
    Segment ka = (Segment)__builtin_amdgcn_kernarg_segment_ptr();
#define TILE_ARGS()                                               \
    do {                                                          \
        ka = (Segment)__builtin_amdgcn_kernarg_segment_ptr();     \
        asm volatile("" : "+s"(ka));                              \
    } while (0)
    TILE_ARGS();
    // entries per query of `seeds` (the u8 instance only runs in the two-pass production mode, never with report_psms > 32)
    const uint32_t kstride = C8 ? WAVE : kp.w.kstride;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const bool w0 = wave == 0;
    const uint32_t n_queued = uni(ka->w.n_deferred[CTR_QUEUED]);
    if (n_queued == 0) return;
    // (the LDS arrays as plain locals: a struct of pointers captured by the lambdas below would live in scratch memory)
    TileLds lds_;
    tile_lds_layout(kp.db.tile_shift, kp.b, &lds_, smem, C8, WING);
    uint32_t* const l_cnt = lds_.cnt;
    uint32_t* const l_bm = lds_.bm;
    // (WING: this workgroup's slice of DevWork::winbuf — global memory, written and read by the wavefronts of one workgroup
    // between workgroup barriers; the compute unit's L1 is theirs alone)
    float* const l_win_lo = WING ? kp.w.winbuf + (size_t)blockIdx.x * 2 * kp.b.fzcap * kp.b.pcap : lds_.win_lo;
    float* const l_win_hi = WING ? l_win_lo + (size_t)kp.b.fzcap * kp.b.pcap : lds_.win_hi;
    uint32_t* const l_hist = lds_.hist;
    uint32_t* const l_sh = lds_.sh;
    uint32_t* const l_pp0 = lds_.pp0;
    uint32_t* const l_pp1 = lds_.pp1;
    uint32_t* const l_pcs = lds_.pcs;
    uint32_t* const l_psum = lds_.psum;
    constexpr uint32_t MARK_BLOCKS = tile_mark_blocks(C8);
    // (constant distances from l_psum — tile_lds_layout lays them out back to back — so that they cost address offsets, not registers)
    uint32_t* const l_nz = l_psum + TILE_WAVES;
    uint32_t* const l_mbits = l_psum + 2 * TILE_WAVES;
    uint32_t* const l_mbase = l_mbits + 4 * MARK_BLOCKS;
    uint16_t* const l_cpid = (uint16_t*)(l_mbase + MARK_BLOCKS);
    uint32_t* const l_qw = lds_.qw;
    const uint32_t TSH = ka->db.tile_shift, TS = 1u << TSH;
    // counter words (SPW slots each) a thread scans: words [tid * wpt, (tid + 1) * wpt) == slots [SPW * tid * wpt, ...), so thread
    // order == slot order.  tile_shift 15, u16: 32 words = eight 16-byte quads per thread.
    const uint32_t wpt = (TS / SPW + TILE_THREADS - 1) / TILE_THREADS;
    const uint32_t ovf_at = (ka->sc.dbg_flags & 16u) ? 2u : 254u;  // (SAGE_HIP_DEBUG_FLAGS=16: tests send every slot with 3+ matches through the overflow path)
    const uint32_t idle_mask = (TS / SPW < TILE_THREADS ? TS / SPW : TILE_THREADS) - 1u;  // (word a thread without a hit adds 0 to)
    const bool fold = ka->sc.min_isotope_err != ka->sc.max_isotope_err;  // scoring.rs:391
    const int isoA = fold ? ka->sc.min_isotope_err : 0, isoB = fold ? ka->sc.max_isotope_err : 0;
    const uint4* __restrict__ frag2 = (const uint4*)ka->db.tm_frag;  // two entries per 16-byte load

    for (uint32_t i = tid; i < TS / SPW; i += TILE_THREADS) l_cnt[i] = 0;  // all-zero between tiles: every scan clears them
    for (uint32_t i = tid; i < TS / 32; i += TILE_THREADS) l_bm[i] = 0;
    // this workgroup's share of the arena: a bump allocator in LDS over chunks taken from the global arena.  Invariant kept by
    // thread 0 between tiles: the chunk has room for a whole tile's candidates, so the wavefronts' allocations during a scan
    // are plain LDS atomics that cannot fail.
    auto refill = [&](uint32_t need) {  // thread 0 only
        const uint32_t cur = l_sh[SH_CHUNK_CUR], lim = l_sh[SH_CHUNK_LIM];
        if (l_sh[SH_ARENA_OK] && lim - cur >= need) return;
        const uint32_t chunk = need > ARENA_CHUNK ? need : ARENA_CHUNK;
        const uint32_t got = atomicAdd(ka->w.arena_ptr, chunk);
        const bool fits = (uint64_t)got + chunk <= ka->w.arena_cap;
        l_sh[SH_CHUNK_CUR] = fits ? got : 0;
        l_sh[SH_CHUNK_LIM] = fits ? got + chunk : 0;
        l_sh[SH_ARENA_OK] = fits ? 1u : 0u;
    };
    if (tid == 0) {  // (the first chunk is taken by the first query that needs one: a retry pass usually has nothing to count)
        l_sh[SH_ARENA_OK] = 0;
        l_sh[SH_CHUNK_CUR] = l_sh[SH_CHUNK_LIM] = 0;
    }

    for (;;) {
        __syncthreads();
        TILE_ARGS();
        if (tid == 0) l_sh[SH_ITEM] = atomicAdd(ka->w.n_deferred + CTR_QUEUE_HEAD, 1u);
        __syncthreads();
        const uint32_t item = uni(l_sh[SH_ITEM]);
        if (item >= n_queued) break;
        const uint32_t spec = uni(ka->w.queue[item]);
        // Retry pass of a two-pass search (DevWork::reuse): the first pass counted this spectrum already and its records —
        // QueryRec, verbatim head, candidate directory in the arena, all of them independent of the trim mode — are still
        // there, in the slot item_of[spec] names.  Only a spectrum whose u8 counters might have wrapped (bit 31) is counted
        // again, in u16, into the same slot.
        uint32_t slot = item;
        if (ka->w.reuse) {
            const uint32_t v = uni(ka->w.item_of[spec]);
            if (!(v >> 31)) continue;
            slot = v & 0x7FFFFFFFu;
        }
        PhaseClock pc;
        pc.start(w0 ? ka->w.dbg : nullptr, item, 2);
        const SpecInfo si = load_spec(kp.sc, kp.b, spec);
        const uint32_t P = si.P;
        const float* __restrict__ masses = ka->b.masses + si.p0;
        const Tol ftol_ = SAGE_LOAD_TOL(ka->sc.fragment_tol);
        for (uint32_t i = tid; i < P; i += TILE_THREADS) {  // database.rs:481 on peak*charge (scoring.rs:360)
            const float m = masses[i];
            for (uint32_t fz = 1; fz <= si.nfz_max; fz++) {
                float lo, hi;
                tol_bounds(ftol_, m * (float)fz, lo, hi);
                l_win_lo[(size_t)(fz - 1) * P + i] = lo;  // stride P: the array index IS the window number fz * P + i
                l_win_hi[(size_t)(fz - 1) * P + i] = hi;
            }
        }
        if (tid < ka->w.qmax) ka->w.qrec[(size_t)slot * ka->w.qmax + tid].potential = 0;  // queries this spectrum does not run
        // IndexedDatabase::query (database.rs:402-425) of EVERY query of the spectrum up front, a wavefront each: three dependent
        // round trips that wavefront 0 used to make query after query while the other seven waited — 18 % of this kernel's time in
        // a wide-window search (three charge states per spectrum, one or two tiles each: profiles/r06_C5_count_phase_clocks.txt)
        const uint32_t n_iso_ = fold ? (uint32_t)(isoB - isoA) + 1u : 1u;
        const uint32_t nq_ = (si.z1 >= si.z0 ? si.z1 - si.z0 + 1u : 0u) * n_iso_;
        const bool windows_ahead = nq_ <= TILE_QW_MAX;
        if (windows_ahead) {
            for (uint32_t q = wave; q < nq_; q += TILE_WAVES) {
                const uint32_t z = si.z0 + q / n_iso_;
                const int iso = isoA + (int)(q % n_iso_);
                const Tol ptol = ka->sc.wide_window ? tol_scaled(si.iso_tol, (float)z) : SAGE_LOAD_TOL(ka->sc.precursor_tol);
                const Window qq = query_window<false>(ka->db.pep_mono, ka->db.np, ptol, si.mzp * (float)z - (float)iso * NEUTRON, ka->db.pep_lut,
                                                      ka->db.pep_lut_bins, ka->db.pep_lut_inv_w);  // scoring.rs:344
                if (lane == 0) *(uint4*)(l_qw + 4 * q) = make_uint4(qq.left, qq.right, qq.first, qq.end);
            }
            __syncthreads();
        }

        for (uint32_t z = si.z0; z <= si.z1; z++) {
            const uint32_t nfz = max_fragment_charge(ka->sc.max_fragment_charge, z) - 1;
            const uint32_t nprobe = P * (nfz < si.nfz_max ? nfz : si.nfz_max);
            const float precursor_mass = si.mzp * (float)z;
            const Tol ptol = ka->sc.wide_window ? tol_scaled(si.iso_tol, (float)z) : SAGE_LOAD_TOL(ka->sc.precursor_tol);
            for (int iso = isoA; iso <= isoB; iso++) {
                TILE_ARGS();
                const size_t qid = (size_t)slot * ka->w.qmax + query_index(kp.sc, si, z, iso);
                // ---- IndexedDatabase::query by wavefront 0, shared through LDS; the query's candidate directory ----
                if (w0) {
                    // (through the position table of the peptide masses, like the narrow kernel since round 4: 3 dependent round trips
                    // instead of ~9 while the other seven wavefronts wait — a third of this kernel's time per spectrum in a wide-window
                    // search, whose three charge-state queries span one or two tiles each: scripts/tile_probe.py wide)
                    Window q;
                    if (windows_ahead) {
                        const uint4 v = *(const uint4*)(l_qw + 4 * query_index(kp.sc, si, z, iso));
                        q.left = v.x; q.right = v.y; q.first = v.z; q.end = v.w;
                    } else {
                        q = query_window<false>(ka->db.pep_mono, ka->db.np, ptol, precursor_mass - (float)iso * NEUTRON, ka->db.pep_lut,
                                                ka->db.pep_lut_bins, ka->db.pep_lut_inv_w);  // scoring.rs:344
                    }
                    if (lane == 0) {
                        l_sh[SH_LEFT] = q.left; l_sh[SH_RIGHT] = q.right; l_sh[SH_FIRST] = q.first; l_sh[SH_END] = q.end;
                        l_sh[SH_MATCHED] = 0; l_sh[SH_SCORED] = 0; l_sh[SH_THR] = 1; l_sh[SH_OVF] = 0; l_sh[SH_NCAND] = 0;
                        const uint32_t lp = q.right < ka->db.np ? q.right : ka->db.np - 1;
                        const uint32_t nt = ka->db.np ? (lp >> TSH) - (q.left >> TSH) + 1 : 1;
                        const uint32_t words = (nt * TILE_WAVES * DIR_WORDS + 3u) & ~3u;
                        refill(words + TS + 8u);
                        uint32_t dir = NONE32;
                        if (l_sh[SH_ARENA_OK]) {
                            dir = l_sh[SH_CHUNK_CUR];
                            l_sh[SH_CHUNK_CUR] = dir + words;
                        } else {
                            atomicAdd(ka->w.n_deferred + CTR_ARENA_OVERFLOW, 1u);
                        }
                        l_sh[SH_DIR] = dir;
                    }
                    l_hist[lane] = 0;
                    for (uint32_t i = lane; i < kstride; i += WAVE) ka->w.seeds[qid * kstride + i] = 0;  // (the scan of any wavefront may overwrite these: the __syncthreads below drains them first)
                }
                __syncthreads();  // also orders win_lo/win_hi and the previous query's reads of sh[]
                const uint32_t left = uni(l_sh[SH_LEFT]), right = uni(l_sh[SH_RIGHT]), first = uni(l_sh[SH_FIRST]), end = uni(l_sh[SH_END]);
                const uint32_t dir = uni(l_sh[SH_DIR]);
                const uint32_t potential = right - left + 1;  // scoring.rs:351
                const uint32_t k = trim_k(potential, ka->sc.report_psms);
                const bool select = potential > k;
                const uint32_t nseed = select ? k : potential;  // slots kept verbatim (<= 64)
                uint32_t acc = 0;
                // a window's bounds (LDS) and its table cells: lut_scale is a power of two, so the products are exact —
                // entries >= lo start at cell floor(lo*scale) and entries <= hi end before cell floor(hi*scale) + 1
                auto probe_bounds = [&](uint32_t pr, float& lo, float& hi) {
                    lo = 1.0f; hi = 0.0f;  // inactive: empty window
                    if (pr < nprobe && first < end) {
                        lo = l_win_lo[pr];
                        hi = l_win_hi[pr];
                    }
                };
                auto probe_cells = [&](float lo, float hi, uint32_t& icl, uint32_t& ich) {
                    lut_cells(lo, hi, ka->db.lut_scale, ka->db.lut_stride, icl, ich);
                };
                pc.mark(0);
                const uint32_t span_fe = end > first ? end - first : 0u;  // (peptides [first, end): `p - first < span_fe`)
                const uint32_t last_pep = right < ka->db.np ? right : ka->db.np - 1;  // slot `right == np` has no peptide behind it
                const uint32_t t0 = left >> TSH, t1 = ka->db.np ? (last_pep >> TSH) : 0;
                // ---- stream: scoring.rs:358-375 over database.rs:480-536 --------------------------------------------------
                // The work of a tile is its RUNS — for every (peak, fragment charge) window the index entries of the tile whose
                // m/z falls into the window's table cells — and run lengths are extremely skewed: fragment masses sit in narrow
                // mass-defect bands, so most windows of a spectrum find nothing while the ones on a band find 100+ entries.
                // One thread per window reads its two table cells; the runs are then flattened into a list of 16-byte cells
                // (two entries each) that ALL threads share evenly, every thread's cells in flight together.  Software
                // pipeline over units (a unit = up to 512 windows of one tile): while tile t's counters are scanned, the
                // entry cells of tile t + 1 and the table reads of tile t + 2 are in flight.
                const uint32_t nb = nprobe ? (nprobe + TILE_THREADS - 1) / TILE_THREADS : 1;  // units per tile (1 up to 512 windows)
                const uint32_t n_units = (t1 - t0 + 1) * nb;
                uint32_t np0 = 0, np1 = 0;  // table values of this thread's window in the NEXT unit (in flight)
                // Quad layout of the table (device_types.h: TM_LUT_LAYOUT == 2), one unit per tile (up to 512 windows — every
                // configuration of BASELINE.json): a thread reads the two table words of its window for FOUR consecutive tiles
                // with two 16-byte loads when the walk enters a quad of tiles, and hands them out tile by tile (x is the tile
                // about to be published; publish rotates).  The table's line requests — most of this kernel's HBM traffic: one
                // line per (window, tile), against ~0.3 for the index entries themselves — fall to a quarter.
                constexpr bool QUAD = TM_LUT_LAYOUT == 2;
                uint4 qa = make_uint4(0u, 0u, 0u, 0u), qb = qa;
                const bool quads = QUAD && nb == 1;
                auto issue_lut = [&](uint32_t u) {
                    if (quads) {
                        const uint32_t t = t0 + u;
                        if (u >= n_units || (u != 0 && (t & 3u) != 0)) return;  // (inside a quad: the words are there already)
                        float lo, hi;
                        uint32_t icl, ich;
                        probe_bounds(tid, lo, hi);
                        probe_cells(lo, hi, icl, ich);
                        qa = qb = make_uint4(0u, 0u, 0u, 0u);
                        if (tid < nprobe && first < end && lo <= hi) {
                            const uint4* __restrict__ lut4 = (const uint4*)ka->db.tm_lut + (size_t)(t >> 2) * ka->db.lut_stride;
                            qa = lut4[icl];
                            qb = lut4[ich];
                        }
                        return;
                    }
                    np0 = np1 = 0;
                    if (u >= n_units) return;
                    const uint32_t t = t0 + u / nb, pr = (u % nb) * TILE_THREADS + tid;
                    float lo, hi;
                    uint32_t icl, ich;
                    probe_bounds(pr, lo, hi);
                    probe_cells(lo, hi, icl, ich);
                    if (pr < nprobe && first < end && lo <= hi) {
                        np0 = ka->db.tm_lut[tm_lut_index(t, icl, ka->db.n_tiles, ka->db.lut_stride)];
                        np1 = ka->db.tm_lut[tm_lut_index(t, ich, ka->db.n_tiles, ka->db.lut_stride)];
                    }
                };
                // cells of the CURRENT unit in flight — named scalars, not arrays (arrays captured by the lambdas below end up in
                // scratch memory): ceN the two entries, cprN the window (thread index of the unit; NONE32: none), cjjN the
                // index of the cell's first entry
#define SAGE_FOR_CELLS(X) X(0) X(1) X(2) X(3)
                static_assert(CELLS_PER_THREAD == 4, "SAGE_FOR_CELLS lists the cells");
#define SAGE_DECL_CELL(I) uint4 ce##I = make_uint4(0u, 0u, 0u, 0u); uint32_t cpr##I = NONE32, cjj##I = 0;
                SAGE_FOR_CELLS(SAGE_DECL_CELL)
#undef SAGE_DECL_CELL
                uint32_t unit_cells = 0;  // cells of the current unit (uniform)
                // locate flattened cell k of the published unit (round 6).  Until then: the owner wavefront by a walk over the eight
                // wave totals, the window by a 6-step search over that wavefront's run starts in LDS — 60 vector instructions and seven
                // dependent LDS reads per cell, 40 % of the instructions of a kernel that is bound by their issue.  A wavefront's 64 cells
                // of one slot are 64 CONSECUTIVE cells, one aligned block of the unit's flattened list: with one bit per cell, set where a
                // run starts, and the rank of the run that owns the block's first cell (published by that run), the owner of cell 64 b + j
                // is base[b] + popcount(bits 1..j of word b) — one uniform LDS read, two v_mbcnt — and `cpid` names its window.
                // (unit u uses the set of mark words u & 1; `mark_lo`: the first flattened cell the marks cover, a multiple of 64 — 0
                // except in a unit that re-marked)
                static_assert(MARK_BLOCKS * 64u >= (C8 ? SAGE_TILE8_CELLS : CELLS_PER_THREAD) * TILE_THREADS, "the marks cover a round of cells");
                static_assert(MARK_BLOCKS * 2u <= TILE_THREADS, "a thread clears one word of the idle set");
                auto locate = [&](uint32_t k, uint32_t mset, uint32_t blk, uint32_t& pr, uint32_t& j) {
                    const uint2 bw = *(const uint2*)(l_mbits + (mset * MARK_BLOCKS + blk) * 2u);  // (one address per wavefront)
                    const uint32_t b_lo = (bw.x >> 1) | (bw.y << 31), b_hi = bw.y >> 1;             // bits 1.. of the word: starts behind the block's first cell
                    const uint32_t r = l_mbase[blk] + __builtin_amdgcn_mbcnt_hi(b_hi, __builtin_amdgcn_mbcnt_lo(b_lo, 0u));
                    pr = l_cpid[r];
                    j = ((l_pp0[pr] >> 1) + (k - l_pcs[pr])) << 1;
                };
                // run-start marks of the windows of the published unit for cells [mark_lo, mark_lo + 64 MARK_BLOCKS): this thread's run
                // is cells [g, g + ncell), the rank-th non-empty run of the unit
                auto mark_run = [&](uint32_t g, uint32_t ncell, uint32_t rank, uint32_t mset, uint32_t mark_lo) {
                    if (ncell == 0u) return;
                    uint32_t* const bits = l_mbits + mset * MARK_BLOCKS * 2u;
                    if (g >= mark_lo && g - mark_lo < MARK_BLOCKS * 64u) atomicOr(&bits[(g - mark_lo) >> 5], 1u << ((g - mark_lo) & 31u));
                    // the blocks whose first cell lies inside the run (most runs: none)
                    const uint32_t b0 = g <= mark_lo ? 0u : (g - mark_lo + 63u) >> 6;
                    const uint32_t last = g + ncell - 1u;  // (>= g)
                    if (last < mark_lo) return;
                    const uint32_t b1 = (last - mark_lo) >> 6;
                    for (uint32_t bb = b0; bb <= b1 && bb < MARK_BLOCKS; bb++) l_mbase[bb] = rank;
                };
#define SAGE_LOAD_CELL(I)                                                        \
    {                                                                            \
        const uint32_t k_ = kbase_ + I * TILE_THREADS + tid;                     \
        cpr##I = NONE32;                                                         \
        ce##I = make_uint4(0u, 0u, 0u, 0u);                                      \
        if (I < CPT && k_ < unit_cells) {                                        \
            locate(k_, mset_, ((kbase_ - mark_lo_) >> 6) + I * TILE_WAVES + wave, cpr##I, cjj##I); \
            ce##I = frag2[cjj##I >> 1]; /* (tm_frag is padded by 2 entries) */   \
        }                                                                        \
    }
// One entry of a cell: inside its run, its window and the precursor window (database.rs:526-533)?  Then its slot's counter goes
// up by one — a RETURNING LDS atomic, branch-free (a lane without a hit adds 0 to a counter word of its own: were the idle
// lanes all sent to one address they would serialise in the LDS), all of a thread's atomics in flight together — so that the statistics the k-select needs are
// kept at hit time: the old count tells which histogram bin the slot leaves and enters, whether it is a new non-empty slot,
// and whether it just reached the tile's pruning threshold (then its bit in the candidate bitmap is set — a count crosses
// the threshold once).  The scan after the tile only has to visit the set bits.
#define SAGE_HIT(I, H, PEP, MZ, JJ)                                                                                  \
    /* (SAGE_HIT_RANGES=1: an empty slot has p0 == p1 and lo > hi; both ranges as one unsigned compare each) */       \
    hit##I##H = SAGE_HIT_RANGES ? ((JJ) - p0_##I < pl_##I && (MZ) >= lo_##I && (MZ) <= hi_##I && (PEP) - first < span_fe) \
                                : (cpr##I != NONE32 && (JJ) >= p0_##I && (JJ) < p1_##I && (MZ) >= lo_##I && (MZ) <= hi_##I && \
                                   (PEP) >= first && (PEP) < end);                                                  \
    x##I##H = hit##I##H ? (PEP) - tb_ : 0u;                                                                          \
    /* no hit: add 0 to a counter word of this thread's own (distinct addresses, no branch, nothing changes) */       \
    old##I##H = atomicAdd(&l_cnt[hit##I##H ? x##I##H >> CSH : tid & idle_mask], hit##I##H ? 1u << ((x##I##H & (SPW - 1u)) * CBITS) : 0u);
// Round 6: the kernel executes vector instructions 93 % of its SIMDs' time, at the full rate of 4 cycles each — and a wavefront whose
// 64 lanes hold NO cell in slot I (a tile's ~900 cells fill 1.8 of the three slots of 512) ran the two branch-free entry tests, their
// idle atomics and their bookkeeping all the same.  `any_I` is wave-uniform: a scalar branch around the slot's work.
#define SAGE_APPLY_CELL(I)                                                      \
    float lo_##I = 1.0f, hi_##I = 0.0f;                                         \
    uint32_t p0_##I = 0, p1_##I = 0;                                            \
    const bool any_##I = __ballot(cpr##I != NONE32) != 0ull;                    \
    bool hit##I##a = false, hit##I##b = false;                                  \
    uint32_t x##I##a = 0, x##I##b = 0, old##I##a = 0, old##I##b = 0;            \
    if (any_##I) {                                                              \
        if (cpr##I != NONE32) {                                                 \
            probe_bounds(pb_ + cpr##I, lo_##I, hi_##I);                         \
            p0_##I = l_pp0[cpr##I];                                             \
            p1_##I = l_pp1[cpr##I];                                             \
        }                                                                       \
        const uint32_t pl_##I = p1_##I - p0_##I;                                \
        SAGE_HIT(I, a, ce##I.x, __uint_as_float(ce##I.y), cjj##I)               \
        SAGE_HIT(I, b, ce##I.z, __uint_as_float(ce##I.w), cjj##I + 1)           \
    }
#define SAGE_ACCOUNT(I, H)                                                                             \
    {                                                                                                  \
        const uint32_t c_ = (old##I##H >> ((x##I##H & (SPW - 1u)) * CBITS)) & CMAX; /* count before this hit */ \
        if (C8 && hit##I##H && c_ >= ovf_at) l_sh[SH_OVF] = 1u; /* (254: the next hit of this slot would carry into its neighbour) */ \
        acc += hit##I##H ? 1u : 0u;                                                                    \
        trans += (hit##I##H && c_ < 3) ? 1u << (c_ * 8) : 0u; /* bytes: 0 -> 1, 1 -> 2, 2 -> 3 */        \
        if (hit##I##H && c_ >= 3 && c_ < HIST_BINS - 1) { /* rare in a search: straight to the histogram (bin 63 = "63 or more") */ \
            atomicSub(&l_hist[c_], 1u);                                                                \
            atomicAdd(&l_hist[c_ + 1], 1u);                                                            \
        }                                                                                              \
        if (hit##I##H && c_ + 1 == thr) atomicOr(&l_bm[x##I##H >> 5], 1u << (x##I##H & 31u));          \
    }
#define SAGE_ACCOUNT_CELL(I) if (any_##I) { SAGE_ACCOUNT(I, a) SAGE_ACCOUNT(I, b) }
                // publish unit u (its table values have arrived in np0 / np1) and put the table reads of unit u + 1 in flight — in two
                // steps with a workgroup barrier between them (the caller's: the one behind the previous unit's hits) and one behind.
                // Step 1: this window's run, the wavefront's totals.  Step 2, when all eight wavefronts' totals are there: where the
                // run lies in the unit's flattened cell list and which of the non-empty runs it is; its marks (locate).  The caller
                // has made sure nobody still reads the previous unit's run table.
                auto publish_totals = [&](uint32_t u) {
                    if (quads) {
                        if (u == 0)  // (a walk that starts inside a quad: bring its first tile to the front)
                            for (uint32_t r = 0; r < (t0 & 3u); r++) {
                                qa = make_uint4(qa.y, qa.z, qa.w, 0u);
                                qb = make_uint4(qb.y, qb.z, qb.w, 0u);
                            }
                        np0 = qa.x;
                        np1 = qb.x;
                        qa = make_uint4(qa.y, qa.z, qa.w, 0u);
                        qb = make_uint4(qb.y, qb.z, qb.w, 0u);
                    }
                    const uint32_t ncell = np1 > np0 ? ((np1 - 1) >> 1) - (np0 >> 1) + 1 : 0;
                    const uint32_t total = wave_sum_dpp(ncell);
                    const uint64_t nzm = __ballot(ncell != 0u);
                    if (lane == 0) {
                        l_psum[wave] = total;
                        l_nz[wave] = (uint32_t)__popcll(nzm);
                    }
                };
                auto publish = [&](uint32_t u) {
                    static_assert(TILE_WAVES == 8, "eight wave totals");
                    // (nothing of step 1 is kept in registers across the barrier: the scan again — 12 instructions against three live values
                    // in a kernel that spills)
                    const uint32_t p0 = np0, p1 = np1;
                    const uint32_t ncell = p1 > p0 ? ((p1 - 1) >> 1) - (p0 >> 1) + 1 : 0;
                    const uint32_t pub_incl = wave_incl_scan_dpp(ncell);  // inclusive prefix over the lanes
                    // exclusive prefixes of the eight wavefronts' totals (cells, non-empty runs), this wavefront's picked by a lane read
                    const uint32_t tc = lane < TILE_WAVES ? l_psum[lane] : 0u, tn = lane < TILE_WAVES ? l_nz[lane] : 0u;
                    const uint32_t ic = wave_incl_scan_dpp(tc), in = wave_incl_scan_dpp(tn);
                    const uint32_t wbase = (uint32_t)__builtin_amdgcn_readlane((int)(ic - tc), (int)uni(wave));
                    const uint32_t nbase = (uint32_t)__builtin_amdgcn_readlane((int)(in - tn), (int)uni(wave));
                    const uint32_t g = wbase + pub_incl - ncell;
                    const uint64_t nzm = __ballot(ncell != 0u);
                    const uint32_t rank = nbase + __builtin_amdgcn_mbcnt_hi((uint32_t)(nzm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nzm, 0u));
                    const uint32_t mset = u & 1u;
                    l_pp0[tid] = p0;
                    l_pp1[tid] = p1;
                    l_pcs[tid] = g;
                    if (ncell) l_cpid[rank] = (uint16_t)tid;
                    mark_run(g, ncell, rank, mset, 0u);
                    if (tid < MARK_BLOCKS * 2u) l_mbits[(mset ^ 1u) * MARK_BLOCKS * 2u + tid] = fresh_zero();  // (the previous unit's: read for the last time two barriers ago)
                    issue_lut(u + 1);
                    lds_barrier();
                    unit_cells = (uint32_t)__builtin_amdgcn_readlane((int)ic, 63);
                    if (pc.slot && tid == 0) {
                        const uint32_t pb = (u % nb) * TILE_THREADS;
                        if (!(ka->sc.dbg_flags & 1024u)) {
                            pc.bytes(DBG_TILE_LUT, 8ull * (nprobe - pb < TILE_THREADS ? nprobe - pb : TILE_THREADS));
                            pc.bytes(DBG_TILE_CELLS, 16ull * unit_cells);
                        }
                    }
                };
                // a unit with more cells than the marks cover (cold): the marks again, from cell `from` on — every thread's run from
                // the run table, between barriers of the whole workgroup (the condition is workgroup-uniform)
                auto remark = [&](uint32_t from, uint32_t mset) {
                    lds_barrier();  // (every wavefront is done with the marks as they are)
                    if (tid < MARK_BLOCKS * 2u) l_mbits[mset * MARK_BLOCKS * 2u + tid] = 0;
                    lds_barrier();
                    const uint32_t p0 = l_pp0[tid], p1 = l_pp1[tid];
                    const uint32_t ncell = p1 > p0 ? ((p1 - 1) >> 1) - (p0 >> 1) + 1 : 0;
                    const uint32_t tn = lane < TILE_WAVES ? l_nz[lane] : 0u;
                    const uint32_t in = wave_incl_scan_dpp(tn);
                    const uint32_t nbase = (uint32_t)__builtin_amdgcn_readlane((int)(in - tn), (int)uni(wave));
                    const uint64_t nzm = __ballot(ncell != 0u);
                    const uint32_t rank = nbase + __builtin_amdgcn_mbcnt_hi((uint32_t)(nzm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nzm, 0u));
                    mark_run(l_pcs[tid], ncell, rank, mset, from);
                    lds_barrier();
                };
                if (tid < MARK_BLOCKS * 4u) l_mbits[tid] = 0;  // (both sets of marks; ordered by the barrier below)
                issue_lut(0);
                publish_totals(0);
                lds_barrier();
                publish(0);
                {
                    const uint32_t kbase_ = 0, mset_ = 0, mark_lo_ = 0;
                    SAGE_FOR_CELLS(SAGE_LOAD_CELL)
                }
                uint32_t trans = 0, t01 = 0, t12 = 0, t23 = 0;  // slots this thread moved from count 0 -> 1, 1 -> 2, 2 -> 3 (histogram transitions)
                for (uint32_t u = 0; u < n_units; u++) {
                    TILE_ARGS();
                    const uint32_t t = t0 + u / nb;
                    const uint32_t tb = t << TSH;
                    // the tile's pruning threshold (wavefront 0 derived it from the histogram of all EARLIER tiles, see below):
                    // a slot whose final count is below it cannot enter the k-select (heap.rs:22 rejects it on arrival)
                    const uint32_t thr = uni(l_sh[SH_THR]);
                    if (pc.slot) {  // (profiling builds of the numbers only: separate the wait for the cells in flight from the apply)
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        pc.mark(7);
                    }
                    {
                        const uint32_t tb_ = tb, pb_ = (u % nb) * TILE_THREADS;
                        {
                            SAGE_FOR_CELLS(SAGE_APPLY_CELL)      // eight returning LDS atomics in flight ...
                            SAGE_FOR_CELLS(SAGE_ACCOUNT_CELL)    // ... then the bookkeeping of the hits
                            t01 += trans & 0xFFu; t12 += (trans >> 8) & 0xFFu; t23 += (trans >> 16) & 0xFFu; trans = 0;
                        }
                        uint32_t mark_lo_ = 0;
                        const uint32_t mset_ = u & 1u;
                        for (uint32_t kbase_ = CPT * TILE_THREADS; kbase_ < unit_cells; kbase_ += CPT * TILE_THREADS) {
                            // (SAGE_HIP_DEBUG_FLAGS=2048: tests let the marks cover one round only, so that every further round re-marks)
                            const uint32_t cover = (ka->sc.dbg_flags & 2048u) ? CPT * TILE_THREADS : MARK_BLOCKS * 64u;
                            if (kbase_ + CPT * TILE_THREADS > mark_lo_ + cover) {  // (workgroup-uniform)
                                remark(kbase_, mset_);
                                mark_lo_ = kbase_;
                            }
                            SAGE_FOR_CELLS(SAGE_LOAD_CELL)  // (a unit with more cells than fit in flight: the rest synchronously)
                            SAGE_FOR_CELLS(SAGE_APPLY_CELL)
                            SAGE_FOR_CELLS(SAGE_ACCOUNT_CELL)
                            t01 += trans & 0xFFu; t12 += (trans >> 8) & 0xFFu; t23 += (trans >> 16) & 0xFFu; trans = 0;
                        }
                    }
                    const bool last_unit = (u % nb) == nb - 1;
                    if (last_unit) {
                        // histogram transitions of this wavefront, one LDS atomic per bin: a slot going c -> c + 1 leaves bin c
                        // (c >= 1) and enters bin c + 1; new non-empty slots count towards scored_candidates
                        const uint32_t s01 = wave_sum_dpp(t01), s12 = wave_sum_dpp(t12), s23 = wave_sum_dpp(t23);
                        if (lane == 0) {
                            if (s01) atomicAdd(&l_sh[SH_SCORED], s01);
                            if (s01 != s12) atomicAdd(&l_hist[1], s01 - s12);
                            if (s12 != s23) atomicAdd(&l_hist[2], s12 - s23);
                            if (s23) atomicAdd(&l_hist[3], s23);
                        }
                        t01 = t12 = t23 = 0;
                    }
                    if (u + 1 < n_units) publish_totals(u + 1);
                    pc.mark(1);
                    lds_barrier();  // every hit of the unit is counted; its run table is free; the next unit's wave totals are there
                    pc.mark(2);
                    TILE_ARGS();
                    if (u + 1 < n_units) {
                        publish(u + 1);
                        const uint32_t kbase_ = 0, mset_ = (u + 1) & 1u, mark_lo_ = 0;
                        SAGE_FOR_CELLS(SAGE_LOAD_CELL)
                    }
                    if (!last_unit) continue;  // (more windows of this tile to come)
                    pc.mark(3);
                    TILE_ARGS();
                    // ---- the NEXT tile's threshold, by wavefront 0: the histogram now holds every slot up to this tile and
                    //      stays put until the next tile's hits (two barriers away).  The k-th largest count so far is a lower
                    //      bound of the heap minimum for every later slot.
                    if (w0) {
                        // slots with count >= lane: the suffix sum as total - inclusive prefix + own (DPP scan: no LDS-crossbar trips —
                        // six dependent ds_bpermute of ~100 cycles each stood here, once per tile, on the wavefront the other seven
                        // wait for at the next barrier)
                        const uint32_t hown = l_hist[lane];
                        const uint32_t hincl = wave_incl_scan_dpp(hown);
                        const uint32_t suffix = (uint32_t)__builtin_amdgcn_readlane((int)hincl, 63) - hincl + hown;
                        const uint64_t ok = __ballot(lane >= 1 && suffix >= k);
                        const uint32_t hmin = ok ? 63u - (uint32_t)__clzll((long long)ok) : 0u;
                        if (lane == 0) l_sh[SH_THR] = hmin > 1 ? hmin : 1;
                    }
                    // ---- scan: only the candidate bits.  Thread tid owns slots [tid * spt, (tid + 1) * spt): lane order == slot
                    //      order, a wavefront owns one contiguous slot range, so the candidates of a wavefront, ranked through a
                    //      prefix sum of popcounts, ARE in slot order.
                    const uint32_t spt = SPW * wpt;               // slots per thread when the window covers the tile (64 at tile_shift 15)
                    // Round 6: a NARROW window inside the tile — a wide-window / DIA query holds ~10^4 candidates, a third of a tile —
                    // used to leave its candidates with the quarter of the threads that own its slots while the rest idled at the
                    // next barrier (20 % of this kernel's time on C5).  Only slots [ws, we) of the tile can have been touched (a hit's
                    // peptide lies in [first, end)), so the threads share THAT range: from vs = ws rounded down to a bitmap word,
                    // spt_v slots each — the smallest power of two whose 512 pieces reach we.  Thread order is still slot order, a
                    // wavefront still owns one contiguous slot range; counters and bits outside the range are clear as they are.
                    const uint32_t ws = first > tb ? first - tb : 0u;
                    const uint32_t we = end < tb + TS ? (end > tb ? end - tb : 0u) : TS;
                    const uint32_t vs = ws & ~31u;
                    const uint32_t span_v = we > vs ? we - vs : 0u;
                    uint32_t spt_v = spt;
                    while (spt_v > SPW && (spt_v >> 1) * TILE_THREADS >= span_v) spt_v >>= 1;
                    const uint32_t x0 = vs + tid * spt_v;  // this thread's first slot (a multiple of spt_v: bits never straddle a word)
                    // (a window that covers the tile — every tile but the two ends of an open search's window — keeps round 5's forms:
                    // thread tid owns [tid * spt, (tid + 1) * spt), one 8-byte read, 16-byte clears)
#ifndef SAGE_TILE_WHOLE_FAST
#define SAGE_TILE_WHOLE_FAST 1
#endif
                    const bool whole_tile = SAGE_TILE_WHOLE_FAST && vs == 0u && spt_v == spt;
                    uint64_t mask = 0;
                    if (whole_tile) {
                        if (tid * spt < TS) {
                            if (spt >= 64) {
                                const uint2 mw = *(const uint2*)(l_bm + tid * 2);
                                mask = ((uint64_t)mw.y << 32) | mw.x;
                            } else if (spt == 32) {
                                mask = l_bm[tid];
                            } else {
                                mask = (l_bm[(tid * spt) >> 5] >> ((tid * spt) & 31u)) & ((1u << spt) - 1u);
                            }
                        }
                    } else if (x0 < TS) {
                        if (spt_v >= 64) {
                            const uint32_t wlo = l_bm[x0 >> 5], whi = (x0 >> 5) + 1 < TS / 32 ? l_bm[(x0 >> 5) + 1] : 0u;
                            mask = ((uint64_t)whi << 32) | wlo;
                        } else if (spt_v == 32) {
                            mask = l_bm[x0 >> 5];
                        } else {
                            mask = (l_bm[x0 >> 5] >> (x0 & 31u)) & ((1u << spt_v) - 1u);
                        }
                    }
                    const uint32_t mine = (uint32_t)__popcll(mask);
                    // (past the first tiles of a window the pruning threshold leaves most wavefronts of most tiles without a single
                    // candidate: no prefix sum then — SAGE_SCAN_SKIP)
                    uint32_t incl = mine;
                    if (!SAGE_SCAN_SKIP || __ballot(mine != 0u) != 0ull) incl = wave_incl_scan_dpp(mine);
                    const uint32_t excl = incl - mine;
                    const uint32_t wave_total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    // this wavefront's run of candidates: one LDS atomic on the workgroup's bump pointer (room for a whole tile
                    // is guaranteed, see refill), one directory entry per (tile, wavefront) — also when the run is empty
                    uint32_t run_at = NONE32;
                    if (dir != NONE32) {
                        uint32_t at = 0;
                        if (lane == 0) {
                            uint32_t n_out = wave_total;
                            if (wave_total) {
                                if (l_sh[SH_ARENA_OK]) {
                                    at = atomicAdd(&l_sh[SH_CHUNK_CUR], (wave_total + 3u) & ~3u);
                                } else {  // the global arena ran out between two tiles: candidates are dropped, the host is told
                                    atomicAdd(ka->w.n_deferred + CTR_ARENA_OVERFLOW, 1u);
                                    at = NONE32;
                                    n_out = 0;
                                }
                            }
                            const uint32_t d = dir + ((t - t0) * TILE_WAVES + wave) * DIR_WORDS;
                            ka->w.arena[d] = at;
                            ka->w.arena[d + 1] = n_out;
                            if (n_out) atomicAdd(&l_sh[SH_NCAND], n_out);
                            if (ka->w.dbg) atomicAdd(ka->w.dbg + (size_t)(item % DBG_BLOCKS) * 32 + DBG_TILE_CAND, 4ull * n_out + 8ull);
                        }
                        run_at = uni(at);
                    }
                    // every lane writes ITS candidates (the set bits of its mask, in slot order) at run position excl + j: lane
                    // order == slot order, so the run is in slot order.  The counters of the first four are read together.
                    if (mine && run_at != NONE32) {
                        const uint32_t x_base = x0;
                        uint64_t m = mask;
                        uint32_t pos = run_at + excl;
                        uint32_t xs[4], cs[4];
#pragma unroll
                        for (uint32_t i = 0; i < 4; i++) {
                            xs[i] = NONE32;
                            cs[i] = 0;
                            if (m) {
                                xs[i] = x_base + (uint32_t)__ffsll((long long)m) - 1;
                                m &= m - 1;
                                cs[i] = l_cnt[xs[i] >> CSH];
                            }
                        }
#pragma unroll
                        for (uint32_t i = 0; i < 4; i++) {
                            if (xs[i] == NONE32) continue;
                            const uint32_t c = (cs[i] >> ((xs[i] & (SPW - 1u)) * CBITS)) & CMAX;
                            const uint64_t g = (uint64_t)tb + xs[i] - left;  // candidate slot
                            // (the verbatim head of the window is written below from the counters; here it leaves a hole)
                            ka->w.arena[pos++] = g < nseed ? 0u : (c << 16) | xs[i];
                        }
                        while (m) {
                            const uint32_t x = x_base + (uint32_t)__ffsll((long long)m) - 1;
                            m &= m - 1;
                            const uint32_t c = (l_cnt[x >> CSH] >> ((x & (SPW - 1u)) * CBITS)) & CMAX;
                            const uint64_t g = (uint64_t)tb + x - left;
                            ka->w.arena[pos++] = g < nseed ? 0u : (c << 16) | x;
                        }
                    }
                    // the first min(k, potential) slots of the window go to the k-select verbatim, whatever their count
                    if ((uint64_t)tb < (uint64_t)left + nseed && (uint64_t)tb + TS > left) {
                        const uint32_t x_lo = x0;
                        for (uint32_t i = 0; i < spt_v; i++) {
                            const uint64_t gx = (uint64_t)tb + x_lo + i;
                            if (gx >= left && gx - left < nseed && x_lo + i < TS) {
                                const uint32_t x = x_lo + i;
                                const uint32_t c = (l_cnt[x >> CSH] >> ((x & (SPW - 1u)) * CBITS)) & CMAX;
                                if (c) ka->w.seeds[qid * kstride + (gx - left)] = (uint16_t)c;
                            }
                        }
                    }
                    pc.mark(4);
                    TILE_ARGS();
                    // clear this thread's counters and candidate bits (its own range only: no other wavefront reads them; what lies
                    // outside [vs, vs + 512 spt_v) was never touched)
                    const uint32_t zr = fresh_zero();  // (not a literal: see fresh_zero)
                    if (whole_tile) {
                        const uint32_t w_lo = tid * wpt;
                        if (w_lo < TS / SPW) {
                            if (wpt >= 4) {
                                for (uint32_t i = 0; i < wpt; i += 4) *(uint4*)(l_cnt + w_lo + i) = make_uint4(zr, zr, zr, zr);
                            } else {
                                for (uint32_t i = 0; i < wpt; i++) l_cnt[w_lo + i] = zr;
                            }
                        }
                        if (tid * spt < TS) {
                            if (spt >= 64) *(uint2*)(l_bm + tid * 2) = make_uint2(zr, zr);
                            else if (((tid * spt) & 31u) == 0) l_bm[(tid * spt) >> 5] = zr;
                        }
                    } else if (x0 < TS) {
                        const uint32_t w_lo = x0 / SPW;                                   // (x0 is a multiple of spt_v >= SPW)
                        const uint32_t w_n = x0 + spt_v <= TS ? spt_v / SPW : (TS - x0) / SPW;  // (the last piece may end with the tile)
                        if (spt_v / SPW >= 4 && w_n == spt_v / SPW) {
                            for (uint32_t i = 0; i < w_n; i += 4) *(uint4*)(l_cnt + w_lo + i) = make_uint4(zr, zr, zr, zr);
                        } else {
                            for (uint32_t i = 0; i < w_n; i++) l_cnt[w_lo + i] = zr;
                        }
                        // (a bitmap word shared by several threads — fewer than 32 slots per thread — belongs to lanes of one
                        // wavefront, which all read their masks above before any of them gets here)
                        if (spt_v >= 32) {
                            for (uint32_t i = 0; i < spt_v / 32 && (x0 >> 5) + i < TS / 32; i++) l_bm[(x0 >> 5) + i] = zr;
                        } else if ((x0 & 31u) == 0) {
                            l_bm[x0 >> 5] = zr;
                        }
                    }
                    pc.mark(5);
                    lds_barrier();  // counters and bits clear again (the candidate stores stay in flight)
                    pc.mark(6);
                    if (tid == 0) refill(TS + 8u);  // room for the next tile's candidates (read again only after two more barriers)
                }
#undef SAGE_LOAD_CELL
#undef SAGE_APPLY_CELL
#undef SAGE_ACCOUNT_CELL
#undef SAGE_ACCOUNT
#undef SAGE_HIT
#undef SAGE_FOR_CELLS
                TILE_ARGS();
                // ---- totals of this query ----
                acc = wave_sum_dpp(acc);
                if (lane == 0 && acc) atomicAdd(&l_sh[SH_MATCHED], acc);
                __syncthreads();
                if (w0) {
                    // the k-th largest count T of the whole window and how many of the slots equal to T (the first ones) do
                    // NOT make the cut: all an order-free trim_hits needs (tile_select_kernel)
                    const uint32_t hmine = l_hist[lane];
                    const uint32_t hincl = wave_incl_scan_dpp(hmine);
                    const uint32_t suffix = (uint32_t)__builtin_amdgcn_readlane((int)hincl, 63) - hincl + hmine;
                    const uint64_t okm = __ballot(lane >= 1 && suffix >= k);
                    const uint32_t T = uni(okm ? 63u - (uint32_t)__clzll((long long)okm) : 0u);
                    const uint32_t n_gt = (uint32_t)__builtin_amdgcn_readlane((int)suffix, (int)(T + 1 < 64 ? T + 1 : 63));
                    const uint32_t n_eq = T ? (uint32_t)__builtin_amdgcn_readlane((int)hmine, (int)(T < 64 ? T : 63)) : 0;
                    const uint32_t big = (uint32_t)__builtin_amdgcn_readlane((int)hmine, 63) != 0;  // some slot matched >= 63 peaks: bins are clipped
                    if (lane == 0) {
                        QueryRec r;
                        r.left = left;
                        r.potential = potential;
                        r.matched = l_sh[SH_MATCHED];
                        r.scored = l_sh[SH_SCORED];
                        r.head = dir;
                        r.z_iso = z | ((uint32_t)(iso + 128) << 8);
                        // bit 0: the heap replay must keep 64-bit keys / no fast select; bit 1: a u8 counter may have overflowed
                        r.pad[0] = big | (l_sh[SH_OVF] ? 2u : 0u) | (T << 8);
                        r.pad[1] = T ? n_eq - (k - n_gt) : 0;             // slots equal to T to skip
                        r.n_dir = dir != NONE32 ? (t1 - t0 + 1) * TILE_WAVES : 0;
                        r.t0 = t0;
                        r.n_cand = l_sh[SH_NCAND];
                        ka->w.qrec[qid] = r;
                    }
                }
            }
        }
    }
}

#undef TILE_ARGS

__global__ __launch_bounds__(TILE_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void tile_count_kernel(const TileParams kp) {
    extern __shared__ __align__(16) unsigned char smem[];
    tile_count_body<false>(kp, smem);
}
__global__ __launch_bounds__(TILE_THREADS) __attribute__((amdgpu_waves_per_eu(6, 6))) void tile_count8_kernel(const TileParams kp) {
    extern __shared__ __align__(16) unsigned char smem[];
    tile_count_body<true>(kp, smem);
}
// Spectra of thousands of peaks (`max_peaks` is the user's: sage-cli input.rs:366): the (peak, fragment charge) windows no longer fit
// the LDS next to a tile's counters — 8 bytes per window, 130 KB for 5 400 peaks x 3 charges — and live in global memory instead.
// Slow next to the LDS instances, and rare.
__global__ __launch_bounds__(TILE_THREADS) __attribute__((amdgpu_waves_per_eu(2, 4))) void tile_count_wing_kernel(const TileParams kp) {
    extern __shared__ __align__(16) unsigned char smem[];
    tile_count_body<false, true>(kp, smem);
}

// A query's candidates are read back through its directory (QueryRec::head / n_dir), defensively: a position outside the
// arena (a count kernel that did not run, a stale record) yields an empty run instead of a wild read.
struct DirRun {
    uint32_t at, n, tile_base;
};
__device__ __forceinline__ DirRun dir_run(const DevWork& w, const QueryRec& rec, uint32_t d) {
    DirRun r{0u, 0u, 0u};
    const uint64_t e = (uint64_t)rec.head + (uint64_t)d * DIR_WORDS;
    if (rec.head == NONE32 || e + DIR_WORDS > w.arena_cap) return r;
    const uint2 v = *(const uint2*)(w.arena + e);
    r.at = v.x;
    r.n = ((uint64_t)v.x + v.y <= w.arena_cap) ? v.y : 0u;
    r.tile_base = (rec.t0 + d / TILE_WAVES) << w.tile_shift;
    return r;
}

// A query's candidate stream by ONE WAVEFRONT, 64 words at a time in slot order, whatever runs they sit in: the runs of a window
// are many and short (one per tile and count-kernel wavefront, a handful of candidates each), and walking them one by one is two
// dependent HBM round trips per run.  Here 64 directory entries are read at once (a lane each), a prefix sum of their lengths
// turns "word k of the block" into (run, offset) — the owner search is six cross-lane reads — and the next 64 words are in
// flight while f(word, tile base of the word's run) works on the current ones.  Holes (words with count 0) are passed on.
// `marks` (round 6): 64 words of LDS of the calling wavefront.  The owner of word k of a block of runs — the last non-empty run
// that starts at or before it — used to be a six-step binary search over the lanes' run starts, six ds_bpermute that depend on each
// other per 64 words; the words of a fetch are consecutive, so every non-empty run that starts inside the fetch marks its first
// word with its lane + 1 and a prefix maximum through DPP (carried from fetch to fetch: starts ascend) names every word's owner —
// two LDS writes and a read instead of the six crossbar trips (tile_select_kernel 3.0 -> ... ms on C5).
template <class F>
__device__ __forceinline__ void for_each_candidate_batch(const DevWork& w, const QueryRec& rec, uint32_t* marks, F&& f) {
    const uint32_t lane = lane_id();
    for (uint32_t d0 = 0; d0 < rec.n_dir; d0 += WAVE) {
        DirRun r{0u, 0u, 0u};
        if (d0 + lane < rec.n_dir) r = dir_run(w, rec, d0 + lane);
        const uint32_t incl = wave_incl_scan_dpp(r.n), excl = incl - r.n;
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        uint32_t carry = 0;  // owner + 1 of the last word of the fetch before (fetches go up in `base`)
        auto fetch = [&](uint32_t base, uint32_t& e, uint32_t& tb) {
            const uint32_t k = base + lane;
            lds_sync();
            marks[lane] = 0u;
            lds_sync();
            if (r.n != 0u && excl - base < WAVE) marks[excl - base] = lane + 1u;  // (unsigned: false for runs that start before `base`)
            lds_sync();
            uint32_t own1 = wave_incl_scan_max_dpp(marks[lane]);
            own1 = own1 > carry ? own1 : carry;
            carry = (uint32_t)__builtin_amdgcn_readlane((int)own1, 63);
            const uint32_t own = own1 ? own1 - 1u : 0u;
            const uint32_t at = (uint32_t)__shfl((int)r.at, (int)own, 64), first = (uint32_t)__shfl((int)excl, (int)own, 64);
            tb = (uint32_t)__shfl((int)r.tile_base, (int)own, 64);
            e = k < total ? w.arena[at + (k - first)] : 0u;
        };
#ifndef SAGE_CAND_FETCH_DEPTH
#define SAGE_CAND_FETCH_DEPTH 3  // fetches of 64 words in flight (round 6: 1 -> 3; the consumers do little per word, a fetch is a round trip to HBM)
#endif
        if (SAGE_CAND_FETCH_DEPTH > 1) {
            // DEPTH fetches in flight, issued in ascending order (the owner carry goes from fetch to fetch), consumed in the same order
            // (named scalars, not arrays: an array captured by `fetch` would live in scratch memory)
#if SAGE_CAND_FETCH_DEPTH == 6
#define SAGE_FETCH_SLOTS(X) X(0) X(1) X(2) X(3) X(4) X(5)
#elif SAGE_CAND_FETCH_DEPTH == 4
#define SAGE_FETCH_SLOTS(X) X(0) X(1) X(2) X(3)
#else
#define SAGE_FETCH_SLOTS(X) X(0) X(1) X(2)
#endif
            constexpr uint32_t DEPTH = SAGE_CAND_FETCH_DEPTH == 6 ? 6 : SAGE_CAND_FETCH_DEPTH == 4 ? 4 : 3;
#define SAGE_FETCH_FIRST(I) uint32_t e##I = 0, t##I = 0; if (I * WAVE < total) fetch(I * WAVE, e##I, t##I);
            SAGE_FETCH_SLOTS(SAGE_FETCH_FIRST)
#undef SAGE_FETCH_FIRST
            for (uint32_t base = 0; base < total; base += DEPTH * WAVE) {
#define SAGE_FETCH_TURN(I)                                                                        \
    if (base + I * WAVE < total) {                                                                \
        f(e##I, t##I);                                                                            \
        if (base + (DEPTH + I) * WAVE < total) fetch(base + (DEPTH + I) * WAVE, e##I, t##I);      \
    }
                SAGE_FETCH_SLOTS(SAGE_FETCH_TURN)
#undef SAGE_FETCH_TURN
            }
#undef SAGE_FETCH_SLOTS
        } else {
            uint32_t e = 0, tb = 0, e_next = 0, tb_next = 0;
            if (total) fetch(0, e, tb);
            for (uint32_t base = 0; base < total; base += WAVE) {
                if (base + WAVE < total) fetch(base + WAVE, e_next, tb_next);
                f(e, tb);
                e = e_next;
                tb = tb_next;
            }
        }
    }
}

// strided sift_down (heap.rs:40-60): element i of this lane's heap lives at hp[i * 64]
template <typename K>
__device__ __forceinline__ void sift_down_strided(K* hp, uint32_t len, uint32_t index, K moving) {
    for (;;) {
        const uint32_t l = index * 2 + 1;
        if (l >= len) break;
        const uint32_t r = l + 1;
        const K vl = hp[l * 64];
        const K vr = r < len ? hp[r * 64] : (K)~(K)0;
        uint32_t smallest = index;
        K sv = moving;
        if (vl < sv) { smallest = l; sv = vl; }
        if (r < len && vr < sv) { smallest = r; sv = vr; }
        if (smallest == index) break;
        hp[index * 64] = sv;  // slice.swap(smallest, index)
        index = smallest;
    }
    hp[index * 64] = moving;
}

// Replace the root by `v` and sift it down (heap.rs:24-25 + :40-60) in ONE LDS round trip.  sift_down always
// descends to the smaller child (the left one on a tie) — a path that does not depend on the value being sifted —
// so with one bit per internal node ("the right child is the smaller one") the whole root-to-leaf path is known
// from registers, its values and their siblings are fetched together, and `v` stops at the first path value
// that is not smaller (values along a heap path never decrease).
constexpr uint32_t HEAP_LEVELS = 6;  // k <= 64
template <typename K>
__device__ __forceinline__ void replace_root_path(K* hp, uint32_t k, uint32_t& rightmin, K v, bool active) {
    const K INF = (K)~(K)0;
    uint32_t p[HEAP_LEVELS + 1];
    K pv[HEAP_LEVELS + 2], sv[HEAP_LEVELS + 1];
    bool valid[HEAP_LEVELS + 1];
    p[0] = 0;
#pragma unroll
    for (uint32_t j = 0; j < HEAP_LEVELS; j++) {
        const uint32_t l = 2 * p[j] + 1;
        valid[j + 1] = (j == 0 ? active : valid[j]) && l < k;
        p[j + 1] = valid[j + 1] ? l + ((rightmin >> p[j]) & 1u) : 0;
    }
#pragma unroll
    for (uint32_t j = 1; j <= HEAP_LEVELS; j++) {
        const uint32_t sib = (p[j] & 1u) ? p[j] + 1 : p[j] - 1;  // the other child of p[j-1]
        pv[j] = valid[j] ? hp[p[j] * 64] : INF;
        sv[j] = (valid[j] && sib < k) ? hp[sib * 64] : INF;
    }
    pv[HEAP_LEVELS + 1] = INF;
    uint32_t d = 0;
#pragma unroll
    for (uint32_t j = 1; j <= HEAP_LEVELS; j++)
        if (valid[j] && d == j - 1 && pv[j] < v) d = j;
    if (!active) return;
#pragma unroll
    for (uint32_t j = 0; j <= HEAP_LEVELS; j++) {
        if (j < d) {
            hp[p[j] * 64] = pv[j + 1];  // slice.swap(smallest, index), one level at a time
            const K nv = j + 1 < d ? pv[j + 2] : v;  // what ends up at p[j+1]
            const bool child_is_left = (p[j + 1] & 1u) != 0;
            const K left = child_is_left ? nv : sv[j + 1], right = child_is_left ? sv[j + 1] : nv;
            const uint32_t bit = 1u << p[j];
            rightmin = right < left ? (rightmin | bit) : (rightmin & ~bit);
        } else if (j == d) {
            hp[p[j] * 64] = v;
        }
    }
}

// Heap keys.  K = u64: the packed PreScore itself.  K = u32: within ONE query charge and isotope error are constant, so
// PreScore's order is (matched, peptide) = (matched, candidate slot): `matched << 21 | slot`, EMPTY == 0.  Valid when the
// window has at most 2^21 slots and every count is below 63 (QueryRec.pad[0] == 0).
constexpr uint32_t K32_SLOT_BITS = 21;
template <typename K> struct ReplayKey;
template <> struct ReplayKey<uint64_t> {
    static __device__ __forceinline__ uint64_t make(uint32_t c, uint32_t slot, uint32_t left, uint32_t z, int iso) {
        return c ? pack_prescore(c, left + slot, z, iso) : PRESCORE_EMPTY;
    }
    static __device__ __forceinline__ uint64_t unpack(uint64_t v, uint32_t, uint32_t, int) { return v; }
    static __device__ __forceinline__ uint32_t matched(uint64_t v) { return prescore_matched(v); }
};
template <> struct ReplayKey<uint32_t> {
    static __device__ __forceinline__ uint32_t make(uint32_t c, uint32_t slot, uint32_t, uint32_t, int) {
        return c ? (c << K32_SLOT_BITS) | slot : 0u;
    }
    static __device__ __forceinline__ uint64_t unpack(uint32_t v, uint32_t left, uint32_t z, int iso) {
        return v ? pack_prescore(v >> K32_SLOT_BITS, left + (v & ((1u << K32_SLOT_BITS) - 1u)), z, iso) : PRESCORE_EMPTY;
    }
    static __device__ __forceinline__ uint32_t matched(uint32_t v) { return v >> K32_SLOT_BITS; }
};

// one lane = one query; every round each lane looks at ONE candidate entry (fetched four at a time) and the lanes whose entry
// can enter — heap.rs:22: later slots have larger peptide indices, so `slice[i] > slice[0]` <=> count >= the root's count —
// replace their root and sift down together
template <typename K>
__device__ __forceinline__ void replay_queries(K* hp, const DevWork& w, const QueryRec& rec, uint64_t qid, uint32_t k, bool live,
                                               uint32_t z, int iso, PhaseClock& pc) {
    typedef ReplayKey<K> RK;
    uint32_t rightmin = 0;  // bit p: the right child of node p is strictly smaller than the left one
    uint32_t hmin = 0;
    if (live) {
        for (uint32_t i = 0; i < k; i++) hp[i * 64] = RK::make(w.seeds[qid * w.kstride + i], i, rec.left, z, iso);  // first k slots verbatim
        for (uint32_t i = k / 2; i-- > 0;) sift_down_strided<K>(hp, k, i, hp[i * 64]);                       // heap.rs:13-15
        for (uint32_t p = 0; 2 * p + 2 < k; p++)
            if (hp[(2 * p + 2) * 64] < hp[(2 * p + 1) * 64]) rightmin |= 1u << p;
        hmin = RK::matched(hp[0]);
    }
    // the lane's position in its query's candidate stream: directory entry d, entry j of run (at, n); runs start 16-byte
    // aligned, so entries are fetched as cells of four
    uint32_t d = 0, j = 0, n = 0, at = 0, tb = 0;
    const uint32_t n_dir = live ? rec.n_dir : 0;
    bool done = n_dir == 0;
    uint4 cell = make_uint4(0u, 0u, 0u, 0u);
    pc.mark(0);
    while (__ballot(!done) != 0ull) {
        bool have = false;
        uint32_t e = 0;
        if (!done) {
            if (j == n) {  // next non-empty run (one directory entry per round: lanes stay in step)
                if (d == n_dir) {
                    done = true;
                } else {
                    const DirRun r = dir_run(w, rec, d++);
                    at = r.at; n = r.n; tb = r.tile_base;
                    j = 0;
                }
            } else {
                if ((j & 3u) == 0) cell = *(const uint4*)(w.arena + at + j);
                const uint32_t q = j & 3u;
                e = q == 0 ? cell.x : q == 1 ? cell.y : q == 2 ? cell.z : cell.w;
                j++;
                have = (e >> 16) != 0 && (e >> 16) >= hmin;  // (count 0: a hole)
            }
        }
        if (__ballot(have) == 0ull) continue;
        replace_root_path<K>(hp, k, rightmin, RK::make(e >> 16, tb + (e & 0xFFFFu) - rec.left, rec.left, z, iso), have);
        if (have) hmin = RK::matched(hp[0]);
    }
    pc.mark(1);
    if (live)
        for (uint32_t i = 0; i < k; i++) w.qres[qid * w.kstride + i] = RK::unpack(hp[i * 64], rec.left, z, iso);
}

// query `qid` of the queue (item * qmax + query) -> where its records live: the same place, or — retry pass reusing the first
// pass's counts — the slot the first pass gave the spectrum
__device__ __forceinline__ uint64_t query_slot(const DevWork& w, uint64_t qid) {
    if (!w.reuse) return qid;
    const uint32_t item = (uint32_t)(qid / w.qmax), q = (uint32_t)(qid % w.qmax);
    return (uint64_t)(w.item_of[w.queue[item]] & 0x7FFFFFFFu) * w.qmax + q;
}
__device__ __forceinline__ bool replay_by_wavefront(const QueryRec& rec, uint64_t n_q, uint64_t wave_max);
__device__ __forceinline__ void tile_replay_block(const DevScorer& sc, const DevWork& w, uint64_t* heap, uint64_t n_q, uint32_t blk, uint64_t wave_max) {
    const uint32_t lane = lane_id();
    const uint64_t qid_in = (uint64_t)blk * 64 + lane;
    const uint64_t qid = qid_in < n_q ? query_slot(w, qid_in) : qid_in;
    PhaseClock pc;
    pc.start(w.dbg, blk, 3);
    QueryRec rec{};
    if (qid_in < n_q) rec = w.qrec[qid];
    const uint32_t k = trim_k(rec.potential, sc.report_psms);
    bool live = rec.potential > k && rec.matched != 0;  // else no k-select: the assembler takes the slots verbatim
    const uint32_t z = rec.z_iso & 0xFFu;
    const int iso = (int)((rec.z_iso >> 8) & 0xFFu) - 128;
    // order-free mode: the heap is only replayed for queries tile_select_kernel cannot take (clipped histogram)
    if (!sc.exact && !(rec.pad[0] & 1u)) live = false;
    if (live && replay_by_wavefront(rec, n_q, wave_max)) live = false;  // (the wavefront-per-query kernel's)
    if (__ballot(live) == 0ull) return;
    const bool small_keys = !(sc.dbg_flags & 2u) &&  // (SAGE_HIP_DEBUG_FLAGS=2: tests force the 64-bit path)
                            __ballot(live && (rec.potential > (1u << K32_SLOT_BITS) || (rec.pad[0] & 1u) != 0)) == 0ull;
    if (small_keys) replay_queries<uint32_t>((uint32_t*)heap + lane, w, rec, qid, k, live, z, iso, pc);
    else replay_queries<uint64_t>(heap + lane, w, rec, qid, k, live, z, iso, pc);
}
// The grids of the four kernels below are capped (TILE_GRID_CAP) and stride over the device-side count of queued spectra:
// a narrow search queues none, and half a million blocks that only read the counter and leave would cost ~0.3 ms per kernel.
// Both replay flavours are launched; the device-side query count, which the host never sees, and each query's stream length
// decide which of the two takes a query (replay_by_wavefront).
__global__ __launch_bounds__(64) void tile_replay_kernel(DevScorer sc, DevWork w, uint64_t wave_max) {
    __shared__ uint64_t heap[64 * 64];  // heap[i * 64 + lane]: conflict-free whatever i each lane is at
    const uint64_t n_q = (uint64_t)w.n_deferred[CTR_QUEUED] * w.qmax;
    if (n_q <= (wave_max & 0xFFFFFFFFull)) return;
    for (uint64_t blk = blockIdx.x; blk * 64 < n_q; blk += gridDim.x) {
        tile_replay_block(sc, w, heap, n_q, (uint32_t)blk, wave_max);
        __syncthreads();
    }
}

// trim_hits of a large-window query WITHOUT replaying the heap (DevScorer::exact == 0): one wavefront walks the verbatim
// slots and the candidate stream in slot order and keeps every slot above the k-th largest count T, plus the last
// (k - #above) slots equal to T — the same k candidates bounded_min_heapify keeps — in slot order.
__device__ __forceinline__ void tile_select_query(const DevScorer& sc, const DevWork& w, const uint64_t qid, uint32_t* marks) {
    const uint32_t lane = lane_id();
    const QueryRec rec = w.qrec[qid];
    const uint32_t k = trim_k(rec.potential, sc.report_psms);
    if (rec.potential <= k || rec.matched == 0 || (rec.pad[0] & 1u)) return;
    const uint32_t z = rec.z_iso & 0xFFu;
    const int iso = (int)((rec.z_iso >> 8) & 0xFFu) - 128;
    const uint32_t T = rec.pad[0] >> 8, skip_eq = rec.pad[1];
    const uint64_t lt = (1ull << lane) - 1ull;
    uint64_t* out = w.qres + qid * w.kstride;
    uint32_t nsel = 0, eq_seen = 0;
    auto offer = [&](uint32_t c, uint32_t pep) {  // one slot per lane (c == 0: none), lanes in slot order
        const uint64_t eqm = __ballot(T != 0 && c == T);
        const uint32_t eq_idx = eq_seen + (uint32_t)__popcll(eqm & lt);
        const bool take = c > T || (T != 0 && c == T && eq_idx >= skip_eq);
        const uint64_t tm = __ballot(take);
        const uint32_t pos = nsel + (uint32_t)__popcll(tm & lt);
        if (take && pos < k) out[pos] = pack_prescore(c, pep, z, iso);
        nsel += (uint32_t)__popcll(tm);
        eq_seen += (uint32_t)__popcll(eqm);
    };
    offer(lane < k ? w.seeds[qid * w.kstride + lane] : 0u, rec.left + lane);  // the first k slots
    for_each_candidate_batch(w, rec, marks, [&](uint32_t e, uint32_t tile_base) {
        // (a candidate below T can never be taken: skip the wavefront's bookkeeping when the whole batch is below)
        if (__ballot((e >> 16) >= T && e != 0u) == 0ull) return;
        offer(e >> 16, tile_base + (e & 0xFFFFu));
    });
    for (uint32_t i = (nsel < k ? nsel : k) + lane; i < k; i += WAVE) out[i] = PRESCORE_EMPTY;  // fewer than k non-empty slots
}
__global__ __launch_bounds__(64) void tile_select_kernel(DevScorer sc, DevWork w) {
    const uint64_t n_q = (uint64_t)w.n_deferred[CTR_QUEUED] * w.qmax;
    __shared__ uint32_t marks[WAVE];  // (for_each_candidate_batch)
    for (uint64_t qid = blockIdx.x; qid < n_q; qid += gridDim.x) tile_select_query(sc, w, qid, marks);
}

// The same replay with ONE WAVEFRONT per query (heap one element per lane, wh32_* / wh_* above): ~10x more work per
// query than the lane-per-query kernel, but every query proceeds in parallel — the better choice while the batch has
// fewer queries than the GPU has wavefront slots (an open search of ~10^4 spectra, or an exact retry pass).
// Which of the two replay kernels takes a query when both are launched (more queries than `wave_max`): the lane-per-query
// kernel lasts as long as the longest stream among its 64 lanes, one word per round, so streams above LANE_MAX_CAND words go
// to the wavefront-per-query kernel, which skims 64 words per step and only pays for the offers that enter the heap.
// (Round 6: 4096 -> 2048 — the wavefront kernel's offers are 2.5x cheaper since wh32_replace_root, and the two kernels run side by
// side: C5's retry pass 13.7 -> 11.7 ms with 2048, 14.3 / 14.9 ms with 1024 / 512: scripts/experiments/r06_lab/gpu_r6r.sh.  Then the
// path of the root replacement from per-lane masks: 11.5 -> 9.0 ms at 2048, 6.6 ms at 512 .. 64 and with every query by wavefront —
// gpu_r7i.sh / gpu_r7j.sh — and DevWork::replay_split's default became "every query by wavefront": this constant only matters when
// SAGE_HIP_REPLAY_WAVE_MAX asks for the split.)
constexpr uint32_t LANE_MAX_CAND = 2048;  // (SAGE_HIP_REPLAY_LANE_MAX overrides it: the upper 32 bits of the kernels' `wave_max`)
__device__ __forceinline__ bool replay_by_wavefront(const QueryRec& rec, uint64_t n_q, uint64_t wave_max) {
    const uint32_t lane_max = (uint32_t)(wave_max >> 32) ? (uint32_t)(wave_max >> 32) : LANE_MAX_CAND;
    wave_max &= 0xFFFFFFFFull;
    if (wave_max == 0) return false;  // (SAGE_HIP_REPLAY_WAVE_MAX=0: tests force the lane-per-query kernel)
    return n_q <= wave_max || rec.n_cand > lane_max;
}
__device__ __forceinline__ void tile_replay_wave_query(const DevScorer& sc, const DevWork& w, const uint64_t qid_in, uint64_t n_q, uint64_t wave_max,
                                                       uint32_t* marks) {
    const uint32_t lane = lane_id();
    const uint64_t qid = query_slot(w, qid_in);
    const QueryRec rec = w.qrec[qid];
    const uint32_t k = trim_k(rec.potential, sc.report_psms);
    if (rec.potential <= k || rec.matched == 0) return;        // no k-select: the assembler takes the slots verbatim
    if (!sc.exact && !(rec.pad[0] & 1u)) return;               // order-free mode: tile_select_kernel took it
    if (!replay_by_wavefront(rec, n_q, wave_max)) return;      // a short stream among many: the lane-per-query kernel's
    const uint32_t z = rec.z_iso & 0xFFu;
    const int iso = (int)((rec.z_iso >> 8) & 0xFFu) - 128;
    const bool small_keys = !(sc.dbg_flags & 2u) && rec.potential <= (1u << K32_SLOT_BITS) && !(rec.pad[0] & 1u);
    const uint32_t seed_c = lane < k ? w.seeds[qid * w.kstride + lane] : 0u;
    // profiling builds of the numbers only (DevWork::dbg, row of kernel 3): [0] cycles, [1] offers that passed the ballot; the
    // large-window byte counter rows [29] / [30] get the replayed queries / their stream words (SAGE_HIP_DEBUG_FLAGS=1024)
    const long long t_start = w.dbg ? clock64() : 0;
    uint32_t n_offers = 0;
    if (small_keys) {  // keys `matched << 21 | slot`, 0 == empty (ReplayKey<uint32_t>)
        Heap32 hp;
        const HeapPath path = heap_path_of_lane();
        wh32_init(hp, seed_c ? (seed_c << K32_SLOT_BITS) | lane : 0u, k);
        wh32_build(hp, k);
        for_each_candidate_batch(w, rec, marks, [&](uint32_t e, uint32_t tile_base) {
            const uint32_t c = e >> 16;
            const uint32_t v = (c << K32_SLOT_BITS) | (tile_base + (e & 0xFFFFu) - rec.left);
            // in slot order; heap.rs:22 — later slots have larger peptide indices, so a count equal to the root's enters
            uint64_t mask = __ballot(c > 0 && c >= (wh32_get(hp, 0) >> K32_SLOT_BITS));
            n_offers += (uint32_t)__popcll(mask);
            while (mask) {
                const uint32_t bit = (uint32_t)__ffsll((long long)mask) - 1;
                mask &= mask - 1;
                wh32_offer_par(hp, k, (uint32_t)__builtin_amdgcn_readlane((int)v, (int)__builtin_amdgcn_readfirstlane(bit)), path);
            }
        });
        if (lane < k) w.qres[qid * w.kstride + lane] = ReplayKey<uint32_t>::unpack(hp.h, rec.left, z, iso);
    } else {
        WaveHeap h;
        const uint64_t sv = seed_c ? pack_prescore(seed_c, rec.left + lane, z, iso) : PRESCORE_EMPTY;
        h.lo = (uint32_t)sv;
        h.hi = (uint32_t)(sv >> 32);
        wh_build(h, k);
        for_each_candidate_batch(w, rec, marks, [&](uint32_t e, uint32_t tile_base) {
            const uint32_t c = e >> 16;
            const uint64_t v = pack_prescore(c, tile_base + (e & 0xFFFFu), z, iso);
            uint64_t mask = __ballot(c > 0 && c >= prescore_matched(wh_get(h, 0)));
            while (mask) {
                const uint32_t bit = (uint32_t)__ffsll((long long)mask) - 1;
                mask &= mask - 1;
                wh_offer(h, k, lane_value(v, bit));
            }
        });
        if (lane < k) w.qres[qid * w.kstride + lane] = ((uint64_t)h.hi << 32) | h.lo;
    }
    if (w.dbg && lane == 0) {
        unsigned long long* row = w.dbg + (size_t)(qid_in % DBG_BLOCKS) * 32 + 24;
        atomicAdd(row + 0, (unsigned long long)(clock64() - t_start));
        atomicAdd(row + 1, (unsigned long long)n_offers);
        if (sc.dbg_flags & 1024u) {
            atomicAdd(row + 5, 1ull);
            atomicAdd(row + 6, (unsigned long long)rec.n_cand);
        }
    }
}
__global__ __launch_bounds__(64) void tile_replay_wave_kernel(DevScorer sc, DevWork w, uint64_t wave_max) {
    const uint64_t n_q = (uint64_t)w.n_deferred[CTR_QUEUED] * w.qmax;
    __shared__ uint32_t marks[WAVE];  // (for_each_candidate_batch)
    for (uint64_t qid = blockIdx.x; qid < n_q; qid += gridDim.x) tile_replay_wave_query(sc, w, qid, n_q, wave_max, marks);
}

// The replay for k > 64 (report_psms > 32): a wavefront per query, the heap in LDS (lh_build / lh_offer_batch), 64-bit keys.
// (the heap: w.kstride entries of dynamic LDS, or of the workgroup's global workspace — DevWork::hugebuf)
template <bool HUGE>
__global__ __launch_bounds__(64) void tile_replay_big_kernel(DevScorer sc, DevWork w) {
    extern __shared__ __align__(16) unsigned char smem[];
    // (dynamic LDS: 256 bytes of marks for for_each_candidate_batch, then — unless the lists live in the global workspace — the heap;
    // no static LDS: the instance may take a compute unit's whole 160 KB)
    uint32_t* const marks = (uint32_t*)smem;
    uint64_t* const heap = HUGE ? (uint64_t*)(w.hugebuf + (size_t)blockIdx.x * w.huge_stride) : (uint64_t*)(smem + WAVE * 4);
    const uint32_t lane = lane_id();
    const uint64_t n_q = (uint64_t)w.n_deferred[CTR_QUEUED] * w.qmax;
    for (uint64_t qid_in = blockIdx.x; qid_in < n_q; qid_in += gridDim.x) {
        const uint64_t qid = query_slot(w, qid_in);
        const QueryRec rec = w.qrec[qid];
        const uint32_t k = trim_k(rec.potential, sc.report_psms);
        if (rec.potential <= k || rec.matched == 0) continue;  // no k-select: the assembler takes the slots verbatim
        const uint32_t z = rec.z_iso & 0xFFu;
        const int iso = (int)((rec.z_iso >> 8) & 0xFFu) - 128;
        __syncthreads();
        for (uint32_t i = lane; i < k; i += WAVE) {
            const uint32_t c = w.seeds[qid * w.kstride + i];
            heap[i] = c ? pack_prescore(c, rec.left + i, z, iso) : PRESCORE_EMPTY;
        }
        lh_build(heap, k);
        for_each_candidate_batch(w, rec, marks, [&](uint32_t e, uint32_t tile_base) {
            const uint32_t c = e >> 16;
            lh_offer_batch(heap, k, pack_prescore(c, tile_base + (e & 0xFFFFu), z, iso), c > 0);
        });
        wave_sync();
        for (uint32_t i = lane; i < k; i += WAVE) w.qres[qid * w.kstride + i] = heap[i];
    }
}

template <bool BIGK>
__device__ __forceinline__ void tile_assemble_item(const DevScorer& sc, const DevBatchView& b, const DevWork& w, unsigned char* smem,
                                                   const uint32_t item) {
    const uint32_t lane = lane_id();
    const uint32_t spec = w.queue[item];
    const uint32_t slot = w.reuse ? w.item_of[spec] & 0x7FFFFFFFu : item;  // (see query_slot)
    const bool fold = sc.min_isotope_err != sc.max_isotope_err;  // scoring.rs:391
    const int isoA = fold ? sc.min_isotope_err : 0, isoB = fold ? sc.max_isotope_err : 0;
    uint64_t* listA = (uint64_t*)smem;
    uint64_t* listB = listA + (fold ? sc.list_cap : 0);
    const SpecInfo si = load_spec(sc, b, spec);
    UList A, B;
    A.items = listA; A.cap = fold ? sc.list_cap : 0; A.stored = 0; A.len = 0; A.ok = true;
    B.items = listB; B.cap = sc.list_cap; B.stored = 0; B.len = 0; B.ok = true;
    uint32_t tot_matched = 0, tot_scored = 0;
    bool cnt_overflow = false;
    for (uint32_t z = si.z0; z <= si.z1; z++) {
        if (fold) { A.stored = 0; A.len = 0; }
        for (int iso = isoA; iso <= isoB; iso++) {
            const size_t qid = (size_t)slot * w.qmax + query_index(sc, si, z, iso);
            const QueryRec rec = w.qrec[qid];
            const uint32_t k = trim_k(rec.potential, sc.report_psms);
            tot_matched += rec.matched;
            cnt_overflow |= (rec.pad[0] & 2u) != 0;
            UList& target = fold ? A : B;
            if (rec.matched == 0) {  // scoring.rs:376-378: the untrimmed all-default vector
                ulist_append_empties(target, rec.potential, sc.kmax);
                continue;
            }
            tot_scored += rec.scored;
            if (BIGK) {  // k up to 256: the same, 64 entries at a time
                const uint32_t m = rec.potential > k ? k : rec.potential;
                for (uint32_t base = 0; base < m; base += WAVE) {
                    const uint32_t i = base + lane;
                    uint64_t v = PRESCORE_EMPTY;
                    if (i < m && rec.potential > k) v = w.qres[qid * w.kstride + i];
                    else if (i < m) {
                        const uint32_t c = w.seeds[qid * w.kstride + i];
                        if (c) v = pack_prescore(c, rec.left + i, z, iso);
                    }
                    ulist_append(target, v, m - base < WAVE ? m - base : WAVE, sc.kmax);
                }
            } else if (rec.potential > k) {  // trim_hits of this query (scoring.rs:380), replayed by tile_replay_kernel
                ulist_append(target, lane < k ? w.qres[qid * w.kstride + lane] : PRESCORE_EMPTY, k, sc.kmax);
            } else {
                const uint32_t c = lane < rec.potential ? w.seeds[qid * w.kstride + lane] : 0;
                ulist_append(target, c ? pack_prescore(c, rec.left + lane, z, iso) : PRESCORE_EMPTY, rec.potential, sc.kmax);
            }
        }
        if (fold) {  // scoring.rs:405 then `hits +=` at :432 / :450
            if (BIGK) ulist_trim_big(A, sc.report_psms);
            else ulist_trim(A, sc.report_psms, sc.exact != 0 || sc.list_cap > 4 * WAVE);
            wave_sync();
            for (uint32_t base = 0; base < A.stored; base += WAVE) {
                const uint64_t v = base + lane < A.stored ? A.items[base + lane] : PRESCORE_EMPTY;
                const uint32_t nvalid = A.stored - base < WAVE ? A.stored - base : WAVE;
                ulist_append(B, v, nvalid, sc.kmax);
            }
            wave_sync();
        }
    }
    if (BIGK) ulist_trim_big(B, sc.report_psms);
    else ulist_trim(B, sc.report_psms, sc.exact != 0 || sc.list_cap > 4 * WAVE);  // scoring.rs:460
    wave_sync();
    if (cnt_overflow) {  // a u8 counter of the count kernel may have wrapped: the spectrum goes through the retry pass (u16 counters)
        if (lane == 0) {
            w.item_of[spec] |= 0x80000000u;  // ... which must count it again
            w.status[spec] = ST_RETRY;
            w.retry[atomicAdd(w.n_deferred + CTR_RETRY, 1u)] = spec;
        }
        return;
    }
    if (lane == 0) {
        if (!(A.ok && B.ok)) atomicAdd(w.n_deferred + CTR_LIST_OVERFLOW, 1u);
        w.status[spec] = (A.ok && B.ok) ? ST_OK : ST_OVERFLOW;
        w.cand_len[spec] = B.stored;
        w.totals[2 * spec] = tot_matched;
        w.totals[2 * spec + 1] = tot_scored;
    }
    for (uint32_t i = lane; i < B.stored; i += WAVE) w.cand[(size_t)spec * sc.kmax + i] = listB[i];
}
template <bool BIGK, bool HUGE = false>
__global__ __launch_bounds__(64) void tile_assemble_kernel(DevScorer sc, DevBatchView b, DevWork w) {
    extern __shared__ __align__(16) unsigned char smem_[];
    unsigned char* const smem = BIGK && HUGE ? w.hugebuf + (size_t)blockIdx.x * w.huge_stride : smem_;  // (the two lists)
    const uint32_t n_items = w.n_deferred[CTR_QUEUED];
    for (uint32_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        tile_assemble_item<BIGK>(sc, b, w, smem, item);
        __syncthreads();
    }
}

// ---- rescoring -------------------------------------------------------------------------------
// the largest of 64 signed 64-bit values, wave-uniform: the DPP pattern of wave_incl_scan_dpp with `max` for `+` (a lane without a
// source keeps its own value), the result read from lane 63 — twelve DPP moves instead of twelve ds_bpermute round trips
__device__ __forceinline__ long long wave_max_i64(long long v) {
#define SAGE_MAX_STEP(CTRL, ROWS)                                                                        \
    {                                                                                                    \
        const int lo = (int)(unsigned long long)v, hi = (int)((unsigned long long)v >> 32);              \
        const unsigned int tlo = (unsigned int)__builtin_amdgcn_update_dpp(lo, lo, CTRL, ROWS, 0xf, false); \
        const unsigned int thi = (unsigned int)__builtin_amdgcn_update_dpp(hi, hi, CTRL, ROWS, 0xf, false); \
        const long long t = (long long)(((unsigned long long)thi << 32) | tlo);                           \
        v = t > v ? t : v;                                                                               \
    }
    SAGE_MAX_STEP(0x111, 0xf)  // row_shr:1
    SAGE_MAX_STEP(0x112, 0xf)  // row_shr:2
    SAGE_MAX_STEP(0x114, 0xf)  // row_shr:4
    SAGE_MAX_STEP(0x118, 0xf)  // row_shr:8
    SAGE_MAX_STEP(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
    SAGE_MAX_STEP(0x143, 0xc)  // row_bcast:31 into rows 2 and 3
#undef SAGE_MAX_STEP
    const unsigned int rlo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned long long)v, 63);
    const unsigned int rhi = (unsigned int)__builtin_amdgcn_readlane((int)((unsigned long long)v >> 32), 63);
    return (long long)(((unsigned long long)rhi << 32) | rlo);
}
__device__ __forceinline__ double from_order_key64(long long k) {  // inverse of order_key64 (the mapping is an involution)
    return __longlong_as_double(k ^ (long long)(((unsigned long long)(k >> 63)) >> 1));
}
// lnfact (scoring.rs:170-177) from the table the host makes with the same correctly rounded ln (capi.hip: scorer_init; it
// covers every u16 argument and every sum of two)
__device__ __forceinline__ double lnfact_dev(uint32_t n, const double* __restrict__ table, uint32_t table_n) {
    return table[n < table_n ? n : table_n - 1u];
}

// ScoreType::score, scoring.rs:179-201.  `ln_i` = ln((summed_b + 1) as f64 * (summed_y + 1) as f64), correctly rounded
// (cr_log_pair below); OpenMSHyperScore takes its f32 ln_1p here.
__device__ __forceinline__ double hyperscore_arg(const Score& s) { return (double)(s.summed_b + 1.0f) * (double)(s.summed_y + 1.0f); }
__device__ __forceinline__ double hyperscore_dev(int score_type, const Score& s, const double ln_i, const double* table, uint32_t tn) {
    double score;
    if (score_type == 0) {
        score = ln_i + lnfact_dev(s.matched_b, table, tn) + lnfact_dev(s.matched_y, table, tn);
    } else {
        const float si = s.summed_b + s.summed_y;
        score = (double)log1pf(si) + lnfact_dev(s.matched_b, table, tn) + lnfact_dev(s.matched_y, table, tn);
    }
    return __builtin_isfinite(score) ? score : 255.0;
}
// The two logarithms of a rescoring round through ONE inlined copy of cr_log (crlog.h): ln(x) always, ln(lambda) when
// `with_lambda` (a spectrum's first round; scoring.rs:522-523).  Two call sites would be two copies of the code in kernels
// that live at the edge of their register budget (eleven copies of both phases cost rescore_kernel 500 scalar spills).
// ACC == false (the hot first-pass instance of rescore_kernel): the fast phase only; `undecided` is set where it could not
// round (~2^-17 of the arguments) and the spectrum goes through the retry pass, whose kernels carry both phases.
// `spare_lane`: lane 63 holds no candidate (a preliminary list is at most 50 long unless report_psms asks for more): it takes
// ln(lambda) in the SAME pass as the candidates' ln(x) — one trip through the ~170 f64-heavy instructions instead of two.
template <bool ACC>
__device__ __forceinline__ double cr_log_pair(const double x, const double lambda, const bool with_lambda, const bool spare_lane,
                                              double& ln_lambda, bool& undecided) {
    double ln_x = 0.0;
    undecided = false;
    const bool shared = with_lambda && spare_lane;  // (wave-uniform)
    const double arg = shared && lane_id() == 63u ? lambda : x;
#pragma nounroll
    for (int it = with_lambda && !shared ? 0 : 1; it < 2; it++) {
        double y;
        if (ACC) {
            y = cr_log(it ? arg : lambda);
        } else {
            const CrLogArg a = cr_log_reduce(it ? arg : lambda);
            bool decided;
            y = cr_log_fast(a, decided);
            if (a.is_special) y = a.special;
            else if (!decided) undecided = true;
        }
        if (it) ln_x = y;
        else ln_lambda = y;
    }
    if (shared) {
        const long long bits = __double_as_longlong(ln_x);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)bits, 63), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(bits >> 32), 63);
        ln_lambda = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    return ln_x;
}

// Rescoring is split in two phases per candidate chunk so that all 64 lanes stay busy and the
// order-sensitive f32 sums still run in the reference's (kind, index, charge) order:
//   A. every (candidate, ion, fragment charge) item of the chunk is matched in parallel
//      (select_most_intense_peak, spectrum.rs:134-159) -> res[item] = peak index or NONE, in LDS;
//   B. one lane per candidate walks ITS items in order and accumulates (scoring.rs:704-754).

// select_most_intense_peak through a direct-index table over the peak masses (core.h: peak_lut_width / peak_lut_entry /
// select_peak_lut, shared with the host: tests/test_core_emulation.py holds it to select_most_intense_peak)
// Four zero registers made on the spot: a plain make_uint4(0, 0, 0, 0) is hoisted out of the rescoring round loop, SPILLED there
// (rescore_kernel is out of registers) and comes back as a scratch load in front of the store it feeds.
__device__ __forceinline__ uint4 fresh_zero4() {
    uint32_t z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return make_uint4(z, z, z, z);
}
__device__ __forceinline__ void build_peak_lut(uint32_t* plut, float& inv_w, const float* pm, uint32_t P) {
    // plut[b] = number of peaks with mass < b * W in the total order (core.h: peak_lut_entry) — by counting instead of 256
    // binary searches: W is a power of two, so bin(m) = floor(m / W) is exact and `mass < b * W` <=> bin(m) < b; masses below
    // +0.0 in the total order (negative, -0.0, negative NaN) lie before every edge, +inf / NaN behind every edge.  A histogram
    // shifted by one bin, then its inclusive prefix sum.
    const uint32_t lane = lane_id();
    const float w = peak_lut_width(P ? pm[P - 1] : 0.0f);
    inv_w = pow2_reciprocal(w);
    static_assert(PLUT_BINS == 4 * WAVE, "four consecutive bins per lane");
    lds_sync();
    *(uint4*)(plut + 4 * lane) = fresh_zero4();
    lds_sync();
    for (uint32_t i = lane; i < P; i += WAVE) {
        const float m = pm[i];
        if (order_key(m) < 0) {
            atomicAdd(&plut[0], 1u);  // before every edge
        } else if (m == m) {
            const float f = __builtin_floorf(m * inv_w);
            if (f < (float)(PLUT_BINS - 1)) atomicAdd(&plut[(uint32_t)f + 1u], 1u);  // bin f: counted from edge f + 1 on
        }
    }
    lds_sync();
    uint4 h = *(const uint4*)(plut + 4 * lane);
    h.y += h.x;
    h.z += h.y;
    h.w += h.z;
    const uint32_t before = wave_incl_scan_dpp(h.w) - h.w;
    *(uint4*)(plut + 4 * lane) = make_uint4(h.x + before, h.y + before, h.z + before, h.w + before);
}
// The peak-presence bitmap that filters score_candidate's lookups (core.h: peak_bitmap_params / _span / _bin, shared with the
// host so that the CPU suite can test that the filter never drops a match).
#ifndef SAGE_COOP_MIN_HITS
#define SAGE_COOP_MIN_HITS 12
#endif
#ifndef SAGE_COOP_MAX_LANES
#define SAGE_COOP_MAX_LANES 2
#endif
// rescore_kernel: hits in a 64-ion chunk above which the wavefront matches the candidate together — when at most
// COOP_MAX_LANES candidates of the spectrum are that heavy (with many heavy candidates — an open search keeps the 50 best of a
// million — every lane is busy anyway and the lanes work on their own)
constexpr uint32_t COOP_MIN_HITS = SAGE_COOP_MIN_HITS, COOP_MAX_LANES = SAGE_COOP_MAX_LANES;
constexpr uint32_t TILE_GRID_CAP = 32768;
constexpr uint32_t HUGE_GRID = 2048;        // workgroups of a wide-list kernel whose lists live in DevWork::hugebuf (one slice each)
constexpr uint32_t RETRY_GRID_CAP = 8192;   // blocks of the narrow exact retry pass  // blocks of the per-query / per-item kernels of the large-window path
__device__ __forceinline__ void build_peak_bitmap(uint32_t* bm, const float* pm, uint32_t P, const PbmReach& reach) {
    const uint32_t lane = lane_id();
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)bm != 0u) __builtin_trap();  // (pbm_bit: the bitmap at LDS address 0)
    static_assert(PBM_WORDS % (4 * WAVE) == 0, "16-byte stores per lane clear the bitmap");
    const uint4 z4 = fresh_zero4();
#pragma unroll
    for (uint32_t i = 0; i < PBM_WORDS / (4 * WAVE); i++) *(uint4*)(bm + 4 * (i * WAVE + lane)) = z4;
    lds_sync();
    // ONE pass: every peak sets its bins; a peak without a safe reach (a negative or non-finite mass, D above 4 Da — core.h:
    // pbm_peak_reach) sets nothing and switches the filter off for the spectrum below
    bool bad = false;
    for (uint32_t i = lane; i < P; i += WAVE) {
        const float m = pm[i];
        float D;
        if (pbm_peak_reach(reach, m, D)) {
            uint32_t b0, b1;
            pbm_peak_span(m, D, b0, b1);
            for (uint32_t bin = b0; bin <= b1; bin++) atomicOr(&bm[(bin & (PBM_BITS - 1u)) >> 5], 1u << (bin & 31u));
        } else {
            bad = true;
        }
    }
    if (__builtin_expect(__ballot(bad) != 0ull, 0)) {
        lds_sync();
        for (uint32_t i = lane; i < PBM_WORDS; i += WAVE) bm[i] = 0xFFFFFFFFu;
    }
}
// The bit of bin `x` modulo PBM_BITS (x: pbm_index of the ion, halved or divided by three for charges 2 and 3).  The bitmap sits at
// LDS address 0 (the scratch part of the rescoring LDS comes first in every kernel that rescores, none of which has static LDS;
// build_peak_bitmap checks it), so the word's byte offset IS its LDS address: (x >> 3) & ((PBM_WORDS - 1) << 2), ds_read_b32, v_bfe_u32.  spectrum_kernel_prepare refuses
// (at scorer creation) a build in which one of these kernels has static LDS, which would push the bitmap off address 0.
typedef const __attribute__((address_space(3))) uint32_t* LdsWordPtr;
__device__ __forceinline__ uint32_t pbm_bit(uint32_t x) {
    const uint32_t w = *(LdsWordPtr)(uintptr_t)((x >> 3) & ((PBM_WORDS - 1u) << 2));
    return __builtin_amdgcn_ubfe(w, x, 1u);  // (v_bfe_u32 takes the offset modulo 32)
}

// remove_matched_peaks (scoring.rs:598-644) on the LDS copy of the spectrum: drop every peak whose (mass, intensity)
// equals a peak matched by the winner's ions, keep order, re-sum the TIC in order.  One wavefront.
__device__ __forceinline__ void remove_matched_peaks_dev(float* pm, float* pi, uint8_t* rm, uint8_t* rm2, uint32_t& P, float& tic,
                                                         const float* __restrict__ wions, uint32_t w_items, uint32_t wmfc,
                                                         const Tol& fragment_tol) {
    const uint32_t lane = lane_id();
    const uint32_t ptop = pow2_floor(P);
    for (uint32_t i = lane; i < P; i += WAVE) rm[i] = 0;
    __syncthreads();
    for (uint32_t t = lane; t < w_items; t += WAVE) {
        const uint32_t ion = t / (wmfc - 1), charge = t % (wmfc - 1) + 1;
        const int pk = select_most_intense_peak_lockstep(pm, pi, P, ptop, wions[ion] / (float)charge, fragment_tol);
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
        const bool kept = i < P && !rm2[i];
        const float mi = i < P ? pm[i] : 0.f, ii = i < P ? pi[i] : 0.f;
        const uint64_t km = __ballot(kept);
        const uint32_t pos = newP + (uint32_t)__popcll(km & ((1ull << lane) - 1ull));
        __syncthreads();
        if (kept) { pm[pos] = mi; pi[pos] = ii; }
        newP += (uint32_t)__popcll(km);
        __syncthreads();
    }
    P = newP;
    float t = 0.0f;  // total_ion_current = intensities.iter().sum::<f32>(), scoring.rs:643
    if (lane == 0) for (uint32_t i = 0; i < P; i++) t += pi[i];
    tic = __shfl(t, 0, 64);
    __syncthreads();
}

// the fields of Score (scoring.rs:17-30) in declaration order == the order of its derived PartialOrd
struct QuickKey {
    uint32_t peptide, matched_b, matched_y;
    float summed_b, summed_y;
    uint32_t longest_b, longest_y;
    double hyperscore;
    float ppm_difference;
    uint32_t charge;
    int iso;
};
// `a > b` under the derived PartialOrd: the first field that differs decides; a NaN makes the pair unordered (false)
__device__ __forceinline__ bool quick_gt(const QuickKey& a, const QuickKey& b) {
    if (a.peptide != b.peptide) return a.peptide > b.peptide;
    if (a.matched_b != b.matched_b) return a.matched_b > b.matched_b;
    if (a.matched_y != b.matched_y) return a.matched_y > b.matched_y;
    if (!(a.summed_b == b.summed_b)) return a.summed_b > b.summed_b;
    if (!(a.summed_y == b.summed_y)) return a.summed_y > b.summed_y;
    if (a.longest_b != b.longest_b) return a.longest_b > b.longest_b;
    if (a.longest_y != b.longest_y) return a.longest_y > b.longest_y;
    if (!(a.hyperscore == b.hyperscore)) return a.hyperscore > b.hyperscore;
    if (!(a.ppm_difference == b.ppm_difference)) return a.ppm_difference > b.ppm_difference;
    if (a.charge != b.charge) return a.charge > b.charge;
    return a.iso > b.iso;
}

// LDS of the rescoring phase.  Two parts: `scratch` (tables and sort keys of one rescoring round; at offset 0 so that the
// bitmap reads — one per (ion, charge) item — address LDS with an immediate base) and `fixed` (the spectrum's peaks, which the
// chimera loop edits in place, and the staging slots of the Feature records).  In the fused narrow kernel the scratch part
// shares its bytes with the LDS of the preliminary phase, which is over when rescoring starts.
struct RescoreLds {
    uint32_t* pbm;        // [PBM_WORDS] peak presence bitmap
    uint32_t* plut;       // [PLUT_BINS] peak position table
    double* s_sorted;     // [64] hyperscores by rank
    long long* s_key;     // [64] sort keys by lane
    QuickKey* qkeys;      // [64] quick_score only
    uint32_t* hdr;        // [RESCORE_HDR_WORDS] the spectrum's scalars the Feature record needs (RescoreHdr): parked here across
                          //      score_candidates instead of in scalar registers the kernel does not have
    uint2* meta;          // [64] lane i: {pep_info, bits of pep_mono} of candidate i — gathered with the ion offsets, read when the
                          //      record is written
    float* pm;            // [pcap] peak masses
    float* pi;            // [pcap] peak intensities
    uint8_t* rm;          // [pcap] chimera: peak selected by the winner
    uint8_t* rm2;         // [pcap]
    uint32_t* stage;      // [stage_records * 30] Feature records on their way out (written by their lanes, stored by the wavefront)
};
constexpr uint32_t FEATURE_WORDS = sizeof(SageFeature) / 4;
static_assert(sizeof(SageFeature) == 120, "Feature records leave LDS as 30 dwords");
// (none for lists wider than a wavefront: rescore_big_kernel's records leave lane by lane — and 512 of them would be 60 KB)
__host__ __device__ inline uint32_t stage_records(const DevScorer& sc) { return sc.big_path ? 0u : sc.chimera ? 1u : sc.report_psms; }
__host__ __device__ inline size_t rescore_scratch_bytes(bool quick) {
    // (the sort keys of a multi-PSM round live in the bitmap's bytes: carve_rescore)
    static_assert(PBM_WORDS * 4 >= 64 * 8 + 64 * 8, "s_sorted + s_key fit the bitmap");
    static_assert(FAST_TIE_WORDS == PBM_WORDS + PLUT_BINS, "the fast-tie replay stages its counts over pbm + plut, contiguous below");
    return (size_t)PBM_WORDS * 4 + PLUT_BINS * 4 + (quick ? 64 * sizeof(QuickKey) : 0);
}
constexpr uint32_t RESCORE_HDR_WORDS = 8;
enum RescoreHdr { HDR_TIC = 0, HDR_MZP = 1, HDR_RT = 2, HDR_IMS = 3, HDR_FILE = 4, HDR_MATCHED = 5, HDR_SCORED = 6 };
constexpr size_t RESCORE_HEAD_BYTES = RESCORE_HDR_WORDS * 4 + 64 * sizeof(uint2);  // hdr + meta, in front of the peaks
__host__ __device__ inline size_t rescore_fixed_bytes(const DevScorer& sc, const DevBatchView& b) {
    const size_t n = RESCORE_HEAD_BYTES + (size_t)b.pcap * 8 + (((size_t)b.pcap * 2 + 7) & ~(size_t)7) + (size_t)stage_records(sc) * sizeof(SageFeature);
    return (n + 15) & ~(size_t)15;
}
__device__ __forceinline__ RescoreLds carve_rescore(unsigned char* scratch, unsigned char* fixed, const DevBatchView& b) {
    RescoreLds l;
    l.pbm = (uint32_t*)scratch;
    l.plut = l.pbm + PBM_WORDS;
    // s_sorted / s_key are written behind score_candidates of a round that reports several PSMs — never a chimera round, so there
    // is no next round that would read the bitmap again: they take its bytes (1 KB of LDS per wavefront less)
    l.s_sorted = (double*)l.pbm;
    l.s_key = (long long*)(l.s_sorted + 64);
    l.qkeys = (QuickKey*)(l.plut + PLUT_BINS);
    l.hdr = (uint32_t*)fixed;
    l.meta = (uint2*)(fixed + RESCORE_HDR_WORDS * 4);
    l.pm = (float*)(fixed + RESCORE_HEAD_BYTES);
    l.pi = l.pm + b.pcap;
    l.rm = (uint8_t*)(l.pi + b.pcap);
    l.rm2 = l.rm + b.pcap;
    l.stage = (uint32_t*)(fixed + RESCORE_HEAD_BYTES + (size_t)b.pcap * 8 + (((size_t)b.pcap * 2 + 7) & ~(size_t)7));
    return l;
}

// ---- score_candidate (scoring.rs:699-759) of the wavefront's 64 candidates: lane i scores ITS candidate (ion table at
//      db.ions + ion_base, `lm1` ions per kind, `nfz` fragment charges; !valid: none) against the spectrum in LDS (peaks pm / pi,
//      presence bitmap pbm, position table plut) and leaves matched / summed / ppm sum / longest runs in `s` — in the
//      reference's (kind, index, charge) order, in chunks of 64 ions: first every (ion, charge) item of the chunk is tested against the
//      peak-presence bitmap — one hit mask per fragment charge 1..3, four ions per trip so that their LDS reads are
//      in flight together, no division — then only the items whose bin is set go through Tolerance::bounds +
//      select_most_intense_peak (direct-index table) and are accumulated, in item order, so the f32 sums are
//      the reference's.  ~90 % of the items of a candidate match nothing.  (Fragment charges above 3 — precursor
//      charge 5+ — are not filtered.)
//      (The kernel is bound by VALU issue — rocprofv3: ~100 % of a SIMD's issue cycles — so what counts is
//      instructions per item; wave-uniform loops over candidates cost 64x per candidate.  And it sits on a knife's edge of
//      register allocation — 96 VGPRs at five wavefronts per SIMD, 52 dwords spilled: two round-3 edits that each removed
//      work — the four-ion trip below cut from ~112 to ~86 VALU instructions (bitmap at a constant LDS address, no clamp per
//      lookup, 32-bit mask halves); the candidate's first ions requested before the bitmap is built, behind LDS-only
//      barriers — each moved the spills (208 -> 224 bytes of scratch) and made the kernel 8-9 % SLOWER, 3.61 -> 3.9 ms per
//      500 000 C3 spectra; shrinking the live state instead (run_matched_packed, the packed match counts) made it 5.5 %
//      faster.  DESIGN.md 4.3.)
//      A candidate with many hits in a chunk (the true peptide: ~35 of its ~47 ions) would keep its lane busy
//      long after the others are done, so the wavefront takes such a chunk TOGETHER: lane i looks up ion i (all
//      charges), then the matches are accumulated in item order by a wave-uniform loop (two readlanes and a few
//      adds per match) and handed back to the candidate's lane.
__device__ __forceinline__ uint32_t lane_run(uint32_t v, uint32_t src_lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)src_lane); }
__device__ __forceinline__ uint64_t lane_run(uint64_t v, uint32_t src_lane) { return lane_value(v, src_lane); }
// LONG: the two Run states of a candidate in 64-bit registers (ion indices beyond 1023: core.h) — rescore_big_kernel's second
// instance; every other caller keeps the one-register form.  FAST: the short divisions (core.h: div_const_fast) — an instance of
// rescore_kernel the host picks when it has bounded the dividends.
#ifndef SAGE_DENSE_HITS
#define SAGE_DENSE_HITS 1  // 1: the lanes' own hits of a round's last chunk through a dense work list in the bitmap's bytes
#endif
constexpr uint32_t DENSE_CAP = 384;  // items of the dense work list: 2 x 4 + 2 bytes each and a flag bit, inside the bitmap's 4 KB
static_assert(DENSE_CAP % 64 == 0 && DENSE_CAP * 10 + DENSE_CAP / 8 <= PBM_WORDS * 4, "the dense work list fits the bitmap");
template <class PC, bool LONG = false, bool FAST = false>
__device__ __forceinline__ void score_candidates(const DevDbView& db, const DevScorer& sc, const uint32_t* pbm, const uint32_t* plut,
                                                 const float* pm, const float* pi, const uint32_t P, const float inv_w,
                                                 const bool valid, const uint64_t ion_base, const uint32_t lm1, const uint32_t nfz,
                                                 const bool any_fz2, const bool any_fz3, const uint32_t nterm_mask, const bool sym_tol,
                                                 Score& s, PC& pc, const bool have_first = false, const float first0 = 0.f,
                                                 const float first1 = 0.f, const float first2 = 0.f, const float first3 = 0.f,
                                                 const bool dense_ok = false) {
    // (have_first: the candidate's first four ions were requested by the caller, ahead of its LDS table builds)
    const uint32_t lane = lane_id();
    typedef typename std::conditional<LONG, uint64_t, uint32_t>::type RunReg;
    constexpr bool DENSE = !LONG;  // (the tags of the dense work list hold ion indices below 2^15)
    RunReg b_run = 0, y_run = 0;  // (run_matched_packed)
    uint32_t mm = 0;                // matched_b | matched_y << 16 (u16 in the reference)
    const bool scored = valid && lm1 && nfz;
    const float* __restrict__ my = db.ions + ion_base;
    const uint32_t nions = scored ? db.n_kinds * lm1 : 0u;
    for (uint32_t j0 = 0; __ballot(j0 < nions) != 0ull; j0 += 64u) {  // (wave-uniform trip count)
        const bool act = j0 < nions;
        const uint32_t n_here = !act ? 0u : nions - j0 < 64u ? nions - j0 : 64u;
        uint64_t m1 = 0, m2 = 0, m3 = 0;
        if (act) {
            // (the ion table is padded by 8: reading past the candidate's last ion is harmless, those bits are masked below)
            const float* __restrict__ q = my + j0;
            float n0, n1, n2, n3;
            if (have_first && j0 == 0) { n0 = first0; n1 = first1; n2 = first2; n3 = first3; }
            else { n0 = q[0]; n1 = q[1]; n2 = q[2]; n3 = q[3]; }
            for (uint32_t r = 0; r < n_here; r += 4) {
                const float i0 = n0, i1 = n1, i2 = n2, i3 = n3;
                n0 = q[r + 4]; n1 = q[r + 5]; n2 = q[r + 6]; n3 = q[r + 7];  // next trip's ions, in flight under this trip's tests
                // one conversion per ion; the bins of its charge states are integer halves / thirds of it (core.h: pbm_index)
                const uint32_t x0 = pbm_index(i0), x1 = pbm_index(i1), x2 = pbm_index(i2), x3 = pbm_index(i3);
                const uint32_t t1 = pbm_bit(x0) | (pbm_bit(x1) << 1) | (pbm_bit(x2) << 2) | (pbm_bit(x3) << 3);
                m1 |= (uint64_t)t1 << r;
                if (any_fz2) {  // (wave-uniform: some candidate of this spectrum has fragment charge 2)
                    const uint32_t t2 = pbm_bit(x0 >> 1) | (pbm_bit(x1 >> 1) << 1) | (pbm_bit(x2 >> 1) << 2) | (pbm_bit(x3 >> 1) << 3);
                    m2 |= (uint64_t)t2 << r;
                }
                if (any_fz3) {
                    const uint32_t t3 = pbm_bit(__umulhi(x0, 0xAAAAAAABu) >> 1) | (pbm_bit(__umulhi(x1, 0xAAAAAAABu) >> 1) << 1) |
                                        (pbm_bit(__umulhi(x2, 0xAAAAAAABu) >> 1) << 2) | (pbm_bit(__umulhi(x3, 0xAAAAAAABu) >> 1) << 3);
                    m3 |= (uint64_t)t3 << r;
                }
            }
            const uint64_t in_chunk = n_here >= 64u ? ~0ull : (1ull << n_here) - 1ull;
            m1 &= in_chunk;
            m2 = nfz >= 2 ? m2 & in_chunk : 0ull;
            m3 = nfz >= 3 ? m3 & in_chunk : 0ull;
            if (nfz > 3) m1 = m2 = m3 = in_chunk;  // (charges above 3 are not filtered: every ion goes through)
        }
        pc.mark(6);  // (phase clocks: the bitmap filter)
        // ---- chunks with many hits: the whole wavefront on one candidate at a time
        const uint32_t hc = act && nfz <= 3 ? (uint32_t)(__popcll(m1) + __popcll(m2) + __popcll(m3)) : 0u;
        uint64_t bigs = (sc.dbg_flags & 32u) ? 0ull : __ballot(hc > COOP_MIN_HITS);  // (SAGE_HIP_DEBUG_FLAGS=32: tests switch it off)
        if ((uint32_t)__popcll(bigs) > COOP_MAX_LANES && !(sc.dbg_flags & 64u)) bigs = 0ull;  // (64: tests take every heavy lane)
        while (bigs) {
            const uint32_t L = (uint32_t)__ffsll((long long)bigs) - 1;
            bigs &= bigs - 1;
            const uint64_t M1 = lane_value(m1, L), M2 = lane_value(m2, L), M3 = lane_value(m3, L);
            const uint64_t base_L = lane_value((uint64_t)ion_base, L);
            const uint32_t lm1_L = (uint32_t)__builtin_amdgcn_readlane((int)lm1, (int)L);
            const bool on1 = (M1 >> lane) & 1ull, on2 = (M2 >> lane) & 1ull, on3 = (M3 >> lane) & 1ull;
            float it1 = 0.f, it2 = 0.f, it3 = 0.f, tm1 = 0.f, tm2 = 0.f, tm3 = 0.f;
            bool ok1 = false, ok2 = false, ok3 = false;
            if (on1 || on2 || on3) {
                const float ionv = db.ions[base_L + j0 + lane];
#define SAGE_COOP_LOOKUP(C, ON, OK, IT, TM)                                                              \
    if (ON) {                                                                                            \
        const float mz = fragment_mz<FAST>(ionv, (C));                                                   \
        float flo, fhi;                                                                                  \
        tol_bounds_mode<FAST>(sc.fragment_tol, sym_tol, mz, flo, fhi);                                   \
        const int pk = select_peak_lut(pm, pi, P, plut, inv_w, flo, fhi);                                \
        if (pk >= 0) {                                                                                   \
const float peak_mass = pm[pk], peak_intensity = pi[pk];                                     \
OK = true;                                                                                   \
IT = peak_intensity;                                                                         \
TM = peak_intensity * __builtin_fabsf(mz - peak_mass) * 2E6f / (mz + peak_mass);             \
        }                                                                                                \
    }
                SAGE_COOP_LOOKUP(1, on1, ok1, it1, tm1)
                SAGE_COOP_LOOKUP(2, on2, ok2, it2, tm2)
                SAGE_COOP_LOOKUP(3, on3, ok3, it3, tm3)
#undef SAGE_COOP_LOOKUP
            }
            const uint64_t K1 = __ballot(ok1), K2 = __ballot(ok2), K3 = __ballot(ok3);
            // the candidate's accumulators, wave-uniform while its matches are added in (ion, charge) order
            float u_sb = lane_valuef(s.summed_b, L), u_sy = lane_valuef(s.summed_y, L), u_pp = lane_valuef(s.ppm_difference, L);
            uint32_t u_mm = (uint32_t)__builtin_amdgcn_readlane((int)mm, (int)L);
            RunReg u_b = lane_run(b_run, L);
            RunReg u_y = lane_run(y_run, L);
            uint64_t anyK = K1 | K2 | K3;
            while (anyK) {
                const uint32_t bit = (uint32_t)__ffsll((long long)anyK) - 1;
                anyK &= anyK - 1;
                uint32_t kind_i = 0, idx = j0 + bit;
                while (idx >= lm1_L) { idx -= lm1_L; kind_i++; }
                const bool nterm = (nterm_mask >> kind_i) & 1u;
#define SAGE_COOP_ADD(K, IT, TM)                                                   \
    if ((K >> bit) & 1ull) {                                                       \
        const float it = lane_valuef(IT, bit), tm = lane_valuef(TM, bit);          \
        u_pp += tm;                                                                \
        if (nterm) { u_mm += 1u; u_sb += it; run_matched_packed(u_b, idx); }       \
        else       { u_mm += 0x10000u; u_sy += it; run_matched_packed(u_y, idx); } \
    }
                SAGE_COOP_ADD(K1, it1, tm1)
                SAGE_COOP_ADD(K2, it2, tm2)
                SAGE_COOP_ADD(K3, it3, tm3)
#undef SAGE_COOP_ADD
            }
            if (lane == L) {
                s.summed_b = u_sb; s.summed_y = u_sy; s.ppm_difference = u_pp;
                mm = u_mm;
                b_run = u_b; y_run = u_y;
                m1 = m2 = m3 = 0ull;  // done
            }
        }
        pc.mark(7);  // (... the heavy candidates, wavefront-wide)
#if SAGE_DENSE_HITS
        // ---- everybody else, DENSE: the lanes' hits of the LAST chunk of a scoring round no other round follows go through
        //      select_most_intense_peak 64 at a time instead of lane by lane behind the lane with the most hits.  The work list lives in
        //      the bitmap's bytes — dead by now: the filter above was its last reader (no later chunk, and only a chimera search
        //      scores a second round against the same bitmap: `dense_ok` is false there).  Three phases: (A) a lane writes one word per
        //      item — its lane, the ion's place in the chunk, the charge, the (series, index) tag — at its prefix-sum position, in
        //      the reference's (ion, charge) order; (B) item i is fetched and looked up by lane i % 64 (the ions of 64 items in flight
        //      together), intensity and ppm term left in place, a flag per item; (C) a lane adds its matched items up in order — the
        //      same f32 additions in the same order as the walk below.
        if (DENSE && dense_ok && !(sc.dbg_flags & 128u) && __ballot(j0 + 64u < nions) == 0ull && __ballot(act && nfz > 3u) == 0ull) {  // (128: tests take the walk)
            const uint32_t hcnt = act ? (uint32_t)(__popcll(m1) + __popcll(m2) + __popcll(m3)) : 0u;
            const uint32_t incl = wave_incl_scan_dpp(hcnt);
            const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (T == 0u) continue;
            if (T <= ((sc.dbg_flags & 256u) ? 64u : DENSE_CAP)) {  // (256: tests cap the list at 64 items — longer ones take the walk)
                uint32_t* const dW = (uint32_t*)pbm;                 // [DENSE_CAP] the item's word -> the matched peak's intensity
                float* const dB = (float*)(dW + DENSE_CAP);          // [DENSE_CAP] its ppm term
                uint16_t* const dT = (uint16_t*)(dB + DENSE_CAP);    // [DENSE_CAP] ion index | n-terminal series << 15
                uint32_t* const dF = (uint32_t*)(dT + DENSE_CAP);    // [DENSE_CAP / 32] item matched a peak
                uint32_t pos = incl - hcnt;
                uint64_t any = m1 | m2 | m3;
                while (any) {
                    const uint32_t bit = (uint32_t)__ffsll((long long)any) - 1;
                    any &= any - 1;
                    uint32_t kind_i = 0, idx = j0 + bit;
                    while (idx >= lm1) { idx -= lm1; kind_i++; }
                    const uint32_t word = idx | (((nterm_mask >> kind_i) & 1u) << 15) | (bit << 16) | (lane << 24);
                    if ((m1 >> bit) & 1ull) dW[pos++] = word | (1u << 22);
                    if (any_fz2 && ((m2 >> bit) & 1ull)) dW[pos++] = word | (2u << 22);
                    if (any_fz3 && ((m3 >> bit) & 1ull)) dW[pos++] = word | (3u << 22);
                }
                lds_sync();
                for (uint32_t base = 0; base < T; base += WAVE) {
                    const uint32_t i = base + lane;
                    const uint32_t word = i < T ? dW[i] : 0u;
                    // (the ion table of the item's candidate: its lane's ion_base, through the crossbar — every lane takes part)
                    const uint32_t src = (word >> 24) << 2;
                    const uint64_t ib = ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute((int)src, (int)(uint32_t)(ion_base >> 32)) << 32) |
                                        (uint32_t)__builtin_amdgcn_ds_bpermute((int)src, (int)(uint32_t)ion_base);
                    bool ok = false;
                    if (i < T) {
                        const float ionv = db.ions[ib + j0 + ((word >> 16) & 63u)];
                        const uint32_t c = (word >> 22) & 3u;
                        float mz = c == 2u ? ionv * 0.5f : ionv;  // (core.h: fragment_mz — x / 1 and x / 2 bit for bit)
                        if (any_fz3 && c == 3u) mz = fragment_mz<FAST>(ionv, 3u);
                        float flo, fhi;
                        tol_bounds_mode<FAST>(sc.fragment_tol, sym_tol, mz, flo, fhi);
                        const int pk = select_peak_lut(pm, pi, P, plut, inv_w, flo, fhi);
                        if (pk >= 0) {
                            const float peak_mass = pm[pk], peak_intensity = pi[pk];
                            ok = true;
                            dW[i] = __float_as_uint(peak_intensity);
                            dB[i] = peak_intensity * __builtin_fabsf(mz - peak_mass) * 2E6f / (mz + peak_mass);
                            dT[i] = (uint16_t)word;
                        }
                    }
                    const uint64_t K = __ballot(ok);
                    if (lane == 0) *(uint2*)(dF + (base >> 5)) = make_uint2((uint32_t)K, (uint32_t)(K >> 32));
                }
                lds_sync();
#define SAGE_DENSE_ADD(Q)                                          \
    {                                                              \
        const float it = __uint_as_float(dW[Q]), tm = dB[Q];       \
        const uint32_t tag = dT[Q];                                \
        s.ppm_difference += tm;                                    \
        if (tag & 0x8000u) {                                       \
            mm += 1u;                                              \
            s.summed_b += it;                                      \
            run_matched_packed(b_run, tag & 0x7FFFu);              \
        } else {                                                   \
            mm += 0x10000u;                                        \
            s.summed_y += it;                                      \
            run_matched_packed(y_run, tag & 0x7FFFu);              \
        }                                                          \
    }
                for (pos = incl - hcnt; pos < incl; pos++) {
                    if (!((dF[pos >> 5] >> (pos & 31u)) & 1u)) continue;
                    SAGE_DENSE_ADD(pos)
                }
#undef SAGE_DENSE_ADD
                continue;
            }
        }
#endif
        // ---- everybody else: the lane walks its own hits
        uint64_t any = m1 | m2 | m3;
        if (!any) continue;
        uint32_t bit = (uint32_t)__ffsll((long long)any) - 1;
        float ionv = my[j0 + bit];
        while (any) {
            any &= any - 1;
            const uint32_t nbit = any ? (uint32_t)__ffsll((long long)any) - 1 : bit;
            const float nion = my[j0 + nbit];  // (the next item's ion is in flight while this one is matched)
            const uint32_t jj = j0 + bit;
            uint32_t kind_i = 0, idx = jj;
            while (idx >= lm1) { idx -= lm1; kind_i++; }
            for (uint32_t c = 1; c <= nfz; c++) {
                if (c <= 3 && !(((c == 1 ? m1 : c == 2 ? m2 : m3) >> bit) & 1ull)) continue;
                // (x / 1.0 == x, x / 2.0 == x * 0.5 bit for bit: the IEEE division only for charges 3 and up; c is wave-uniform)
                const float mz = fragment_mz<FAST>(ionv, c);
                float flo, fhi;
                tol_bounds_mode<FAST>(sc.fragment_tol, sym_tol, mz, flo, fhi);
                const int pk = select_peak_lut(pm, pi, P, plut, inv_w, flo, fhi);
                if (pk < 0) continue;
                const float peak_mass = pm[pk], peak_intensity = pi[pk];
                s.ppm_difference += peak_intensity * __builtin_fabsf(mz - peak_mass) * 2E6f / (mz + peak_mass);
                if ((nterm_mask >> kind_i) & 1u) {
                    mm += 1u;
                    s.summed_b += peak_intensity;
                    run_matched_packed(b_run, idx);
                } else {
                    mm += 0x10000u;
                    s.summed_y += peak_intensity;
                    run_matched_packed(y_run, idx);
                }
            }
            bit = nbit;
            ionv = nion;
        }
    }
    s.matched_b = mm & 0xFFFFu;
    s.matched_y = mm >> 16;
    s.longest_b = run_longest_packed(b_run);
    s.longest_y = run_longest_packed(y_run);
}

// One Feature record (scoring.rs:504-594) of a scored candidate: `rank_field` is Feature.rank, `next` / `best` the hyperscores of
// the next rank (0 if none) and of rank 0.
__device__ __forceinline__ SageFeature make_feature(const uint32_t info, const float calc, const uint32_t spec_index, const uint32_t pep,
                                                    const uint32_t z, const int iso, const Score& s, const double h, const double next,
                                                    const double best, const uint32_t rank_field, const double lambda, const double ln_lambda, const float mzp,
                                                    const float rt, const float ims, const uint32_t fid, const float tic,
                                                    const uint32_t tot_scored, const double* __restrict__ lnfact_table,
                                                    const uint32_t lnfact_n) {
    const float precursor_mass = mzp * (float)z;
    const uint32_t k = s.matched_b + s.matched_y;
    const double log10_poisson =
        ((double)k * ln_lambda - lambda - lnfact_dev(k, lnfact_table, lnfact_n)) / 2.302585092994046;
    const float isotope_error = (float)iso * NEUTRON;
    const float delta_mass =
        (precursor_mass - calc - isotope_error) * 2E6f / (precursor_mass - isotope_error + calc);
    const uint32_t plen = info & 0xFFFF;
    SageFeature f;
    f.spec_index = spec_index;
    f.peptide_idx = pep;
    f.rank = rank_field;  // scoring.rs:541, 664
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
    f.scored_candidates = tot_scored;
    f.peptide_len = plen;
    f.file_id = fid;
    f.charge = (uint8_t)z;
    f.missed_cleavages = (uint8_t)(info >> 24);
    for (int q = 0; q < 6; q++) f.pad[q] = 0;
    return f;
}

// Scorer::build_features / score_chimera_fast / quick_score (scoring.rs:478-595, 648-672, 255-298) of ONE spectrum by one
// wavefront.  Lane i holds candidate i of the trimmed preliminary list (`mine`, PRESCORE_EMPTY beyond its end); the peaks are in
// R.pm / R.pi already (P of them).
//
// The stable sort of scoring.rs:495 makes the ORDER of the preliminary list observable where equal hyperscores meet at a
// reported rank.  `list_is_exact`: the list is in the reference's heap-layout order (lane order decides ties).  Otherwise it
// came from order-free trims — the right candidates in some other order — and such a tie cannot be settled
// here: the function returns false with nothing final reported (`queue_on_tie`: after queueing the spectrum for the exact
// retry pass; else the caller settles it itself).
// ---- late arguments -------------------------------------------------------------------------------------------------------------
// rescore_spectrum needs ~40 scalar registers' worth of arguments only AFTER score_candidates (result pointers, the factorial
// table, the tie / retry bookkeeping, report_psms ...).  Held in registers across the matching loops they do not fit: round 4's
// rescore_kernel wrote ~90 of them to spill lanes in front of score_candidates and read ~140 back behind it — a tenth of the
// kernel's vector-ALU issue slots, once per spectrum.  The kernel's own arguments never need saving: they sit in the kernarg
// segment, one scalar load away.  LateArgs<K> is how the part of rescore_spectrum behind score_candidates reads them:
//   K = void:            from the references it was handed (kernels whose argument list is not a RescoreKernargs)
//   K = RescoreKernargs: from the kernarg segment, through a pointer the compiler cannot see through (`refresh`) — so the loads
//                        stay where they are written instead of being hoisted to the kernel's entry and spilled.
struct RescoreKernargs {  // THE argument list of rescore_kernel (one struct, so that the segment's layout is this struct's)
    DevDbView db;
    DevScorer sc;
    DevBatchView b;
    DevWork w;
    const double* lnfact_table;
    uint32_t lnfact_n;
    SageFeature* out;
    uint32_t* out_count;
    uint8_t* keep;
};
template <class K>
struct LateArgs {  // K = void
    const DevDbView& db;
    const DevScorer& sc;
    const DevBatchView& b;
    const DevWork& w;
    const double* lnfact_table_;
    uint32_t lnfact_n_;
    SageFeature* out_;
    uint32_t* out_count_;
    __device__ __forceinline__ LateArgs(const DevDbView& db_, const DevScorer& sc_, const DevBatchView& b_, const DevWork& w_, const double* t,
                                        uint32_t tn, SageFeature* o, uint32_t* oc)
        : db(db_), sc(sc_), b(b_), w(w_), lnfact_table_(t), lnfact_n_(tn), out_(o), out_count_(oc) {}
    __device__ __forceinline__ void refresh() {}
    __device__ __forceinline__ const float* ions() const { return db.ions; }
    __device__ __forceinline__ uint32_t report_psms() const { return sc.report_psms; }
    __device__ __forceinline__ uint32_t chimera() const { return sc.chimera; }
    __device__ __forceinline__ uint32_t min_matched_peaks() const { return sc.min_matched_peaks; }
    __device__ __forceinline__ int score_type() const { return sc.score_type; }
    __device__ __forceinline__ uint32_t xcd_chunk() const { return sc.xcd_chunk; }
    __device__ __forceinline__ uint32_t batch_n() const { return b.n; }
    __device__ __forceinline__ uint32_t spec_base() const { return b.spec_base; }
    __device__ __forceinline__ uint32_t* status() const { return w.status; }
    __device__ __forceinline__ uint32_t* retry() const { return w.retry; }
    __device__ __forceinline__ uint32_t* counters() const { return w.n_deferred; }
    __device__ __forceinline__ const uint32_t* cnt_store() const { return w.cnt_store; }
    __device__ __forceinline__ uint32_t cnt_stride() const { return w.cnt_stride; }
    __device__ __forceinline__ const double* lnfact_table() const { return lnfact_table_; }
    __device__ __forceinline__ uint32_t lnfact_n() const { return lnfact_n_; }
    __device__ __forceinline__ SageFeature* out() const { return out_; }
    __device__ __forceinline__ uint32_t* out_count() const { return out_count_; }
    __device__ __forceinline__ uint8_t* keep(uint8_t* k) const { return k; }
};
template <>
struct LateArgs<RescoreKernargs> {
    typedef const __attribute__((address_space(4))) RescoreKernargs* Segment;
    Segment ka;
    __device__ __forceinline__ LateArgs(const DevDbView&, const DevScorer&, const DevBatchView&, const DevWork&, const double*, uint32_t, SageFeature*,
                                        uint32_t*)
        : ka((Segment)__builtin_amdgcn_kernarg_segment_ptr()) {}
    __device__ __forceinline__ void refresh() {
        ka = (Segment)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
    }
    __device__ __forceinline__ const float* ions() const { return ka->db.ions; }
    __device__ __forceinline__ uint32_t report_psms() const { return ka->sc.report_psms; }
    __device__ __forceinline__ uint32_t chimera() const { return ka->sc.chimera; }
    __device__ __forceinline__ uint32_t min_matched_peaks() const { return ka->sc.min_matched_peaks; }
    __device__ __forceinline__ int score_type() const { return ka->sc.score_type; }
    __device__ __forceinline__ uint32_t xcd_chunk() const { return ka->sc.xcd_chunk; }
    __device__ __forceinline__ uint32_t batch_n() const { return ka->b.n; }
    __device__ __forceinline__ uint32_t spec_base() const { return ka->b.spec_base; }
    __device__ __forceinline__ uint32_t* status() const { return ka->w.status; }
    __device__ __forceinline__ uint32_t* retry() const { return ka->w.retry; }
    __device__ __forceinline__ uint32_t* counters() const { return ka->w.n_deferred; }
    __device__ __forceinline__ const uint32_t* cnt_store() const { return ka->w.cnt_store; }
    __device__ __forceinline__ uint32_t cnt_stride() const { return ka->w.cnt_stride; }
    __device__ __forceinline__ const double* lnfact_table() const { return ka->lnfact_table; }
    __device__ __forceinline__ uint32_t lnfact_n() const { return ka->lnfact_n; }
    __device__ __forceinline__ SageFeature* out() const { return ka->out; }
    __device__ __forceinline__ uint32_t* out_count() const { return ka->out_count; }
    __device__ __forceinline__ uint8_t* keep(uint8_t*) const { return ka->keep; }
};

template <bool ACC, class KA, bool FAST = false, class PC>
__device__ __forceinline__ bool rescore_spectrum(const DevDbView& db, const DevScorer& sc, const DevBatchView& b, const DevWork& w,
                                                 const double* __restrict__ lnfact_table, uint32_t lnfact_n,
                                                 SageFeature* __restrict__ out, uint32_t* __restrict__ out_count,
                                                 uint8_t* __restrict__ keep, const RescoreLds& R, const uint32_t spec, uint32_t P,
                                                 const uint64_t mine, const uint32_t tot_matched, const uint32_t tot_scored,
                                                 const bool list_is_exact, const bool queue_on_tie, PC& pc) {
    // keep != nullptr: Scorer::quick_score with prefilter_low_memory (scoring.rs:270-289) instead of build_features
    const uint32_t lane = lane_id();
    uint32_t* const pbm = R.pbm;
    uint32_t* const plut = R.plut;
    float* const pm = R.pm;
    float* const pi = R.pi;
    const uint32_t pep = prescore_peptide(mine);
    const bool valid = pep != 0xFFFFFFFFu;  // scoring.rs:489
    const uint32_t z = prescore_charge(mine);
    const int iso = prescore_iso(mine);
    const uint32_t nfz = max_fragment_charge(sc.max_fragment_charge, z) - 1;
    uint64_t ion_base = 0;
    uint32_t lm1 = 0;
    uint32_t pep_rec = 0;   // the candidate's record and mass: requested here, first looked at behind the table builds below, which
    float pep_mass = 0.0f;  // run on LDS alone while these gathers are on their way
    if (valid) {
        ion_base = db.ion_off[pep];
        pep_rec = db.pep_info[pep];
        pep_mass = db.pep_mono[pep];
    }
    // (registers are what this kernel is short of: what only the Feature record needs waits in LDS — R.meta, R.hdr — and nothing
    // is kept that two instructions recompute)
    if (lane == 0) {
        float ims = 0.0f;
        if (b.ims) { const float v = b.ims[spec]; ims = v == v ? v : 0.0f; }
        static_assert(HDR_TIC == 0 && HDR_MZP == 1 && HDR_RT == 2 && HDR_IMS == 3 && HDR_FILE == 4 && HDR_MATCHED == 5 && HDR_SCORED == 6, "two 16-byte stores");
        *(uint4*)R.hdr = make_uint4(__float_as_uint(b.tic[spec]), __float_as_uint(b.precursor_mz[spec] - PROTON) /* scoring.rs:502 */,
                                    __float_as_uint(b.rt ? b.rt[spec] : 0.0f), __float_as_uint(ims));
        *(uint4*)(R.hdr + 4) = make_uint4(b.file_id ? b.file_id[spec] : 0u, tot_matched, tot_scored, 0u);
    }
#define SAGE_N_ITEMS (valid ? db.n_kinds * lm1 * nfz : 0u)  /* (ion, fragment charge) pairs of this candidate */

    double ln_lambda = 0.0;  // (cr_log_pair, first round)
    lds_sync();  // (the peaks are in LDS; the gathers above stay in flight)
    pc.mark(0);
    const bool sym_tol = (sc.tol_mode & TOL_SYM) != 0u;  // (core.h: TolMode — the host's, once per scorer)
    uint32_t nterm_mask = 0;  // bit k: ion kind k is a / b / c (counts towards matched_b, scoring.rs:727-731)
    for (uint32_t k = 0; k < db.n_kinds; k++) nterm_mask |= (db.ion_kinds[k] <= 2 ? 1u : 0u) << k;
    uint32_t n_emitted = 0;
    const bool any_fz2 = __ballot(valid && nfz >= 2) != 0ull, any_fz3 = __ballot(valid && nfz >= 3) != 0ull;
    for (uint32_t round = 0;; round++) {
        float inv_w;
        build_peak_lut(plut, inv_w, pm, P);
        // (built once: after remove_matched_peaks the bitmap is a superset of the remaining peaks' bins — still conservative)
#ifndef SAGE_ION_PREFETCH
#define SAGE_ION_PREFETCH 0  // 1: the candidates' first ions are requested between the two table builds (A/B: see DESIGN.md 4.3)
#endif
        float first0 = 0.f, first1 = 0.f, first2 = 0.f, first3 = 0.f;
        if (round == 0 && valid) {
            // ions per kind = peptide length - 1 (ion_series.rs:68-85; the ion table holds n_kinds * (L - 1) values per peptide): from
            // the peptide's record, not as (ion_off[pep + 1] - ion_off[pep]) / n_kinds — a 64-bit division per candidate
            const uint32_t plen = pep_rec & 0xFFFFu;
            lm1 = db.n_kinds && plen ? plen - 1u : 0u;
            R.meta[lane] = make_uint2(pep_rec, __float_as_uint(pep_mass));  // (for the Feature record of a reporting lane)
            if (SAGE_ION_PREFETCH && lm1 && nfz) {
                const float* __restrict__ q = db.ions + ion_base;  // (padded by 8: harmless past a short candidate's end)
                first0 = q[0]; first1 = q[1]; first2 = q[2]; first3 = q[3];
            }
        }
        if (round == 0) build_peak_bitmap(pbm, pm, P, sc.pbm_reach);
        lds_sync();
        if (round == 0) {
        if (pc.slot) {  // bytes this spectrum's rescoring asks for: peaks, candidate records, every candidate's ion table
            const uint32_t ion_bytes = wave_sum(valid ? 4u * db.n_kinds * lm1 + 24u : 0u);
            const uint32_t ncand = (uint32_t)__popcll(__ballot(mine != PRESCORE_EMPTY));
            if (lane == 0) pc.bytes(DBG_RESCORE, 8ull * P + 8ull * ncand + ion_bytes);
            // shape of the work: (ion, charge) items of all candidates, of the longest candidate, candidates, peaks
            const uint32_t items = wave_sum(SAGE_N_ITEMS);
            uint32_t longest = SAGE_N_ITEMS;
            for (int o = 32; o; o >>= 1) { const uint32_t v = (uint32_t)__shfl_xor((int)longest, o, 64); longest = v > longest ? v : longest; }
            const uint32_t nvalid = (uint32_t)__popcll(__ballot(valid));
            (void)items; (void)longest; (void)nvalid;  // (slots 5..7 now hold phases)
        }
        }
        Score s;
        s.peptide = 0;  // (the lane's pep / z / iso stand in for the fields of the same name)
        s.precursor_charge = 0;
        s.isotope_error = 0;
        s.matched_b = s.matched_y = 0;
        s.summed_b = s.summed_y = 0.0f;
        s.ppm_difference = 0.0f;
        s.longest_b = s.longest_y = 0;
        pc.mark(5);  // (... the peak table and the bitmap)
        score_candidates<PC, false, FAST>(db, sc, pbm, plut, pm, pi, P, inv_w, valid, ion_base, lm1, nfz, any_fz2, any_fz3, nterm_mask, sym_tol,
                                          s, pc, SAGE_ION_PREFETCH && round == 0, first0, first1, first2, first3, !sc.chimera);
        pc.mark(1);  // (... the lanes' own hits)
        // ---- from here on: the arguments through `la`, the spectrum's scalars from R.hdr (see LateArgs) ----
        LateArgs<KA> la(db, sc, b, w, lnfact_table, lnfact_n, out, out_count);
        la.refresh();
        const double lambda = (double)R.hdr[HDR_MATCHED] / (double)R.hdr[HDR_SCORED];  // scoring.rs:499
        double h = 0.0;
        bool pass = false;
        bool ln_undecided;
        const double ln_i = cr_log_pair<ACC>(hyperscore_arg(s), lambda, round == 0, ((__ballot(valid) >> 63) & 1ull) == 0ull, ln_lambda, ln_undecided);
        if (__builtin_expect(!ACC && __ballot(ln_undecided && (valid || round == 0)) != 0ull, 0)) {
            // (the hot instance carries the logarithm's fast phase only: this spectrum again in the retry pass, like a tie)
            if (queue_on_tie && lane == 0) {
                la.status()[spec] = ST_RETRY;
                la.retry()[atomicAdd(la.counters() + CTR_RETRY, 1u)] = spec;
                la.out_count()[spec] = 0;
            }
            return false;
        }
        if (valid) {
            s.ppm_difference /= s.summed_b + s.summed_y;  // scoring.rs:759
            h = hyperscore_dev(la.score_type(), s, ln_i, la.lnfact_table(), la.lnfact_n());
            pass = (s.matched_b + s.matched_y) >= la.min_matched_peaks();  // scoring.rs:491
        }
        const uint32_t report_psms = la.report_psms();
        const bool chimera = la.chimera() != 0u;
        const uint32_t per_round = chimera ? 1u : report_psms;
        if (keep) {
            // quick_score, prefilter_low_memory (scoring.rs:270-289): bounded_min_heapify(&mut scores, k) keeps the k
            // largest elements.  heap.rs compares with `<` / `>`, i.e. the DERIVED PartialOrd of Score — lexicographic
            // in field order, peptide first (scoring.rs:17-30) — not its hyperscore Ord.  The set of the k largest does
            // not depend on the heap's internal order, so rank each passing candidate by that order directly.
            QuickKey* qk = R.qkeys;
            QuickKey mk;
            mk.peptide = pep; mk.matched_b = s.matched_b; mk.matched_y = s.matched_y;
            mk.summed_b = s.summed_b; mk.summed_y = s.summed_y; mk.longest_b = s.longest_b; mk.longest_y = s.longest_y;
            mk.hyperscore = h; mk.ppm_difference = s.ppm_difference; mk.charge = z; mk.iso = iso;
            qk[lane] = mk;
            const uint64_t pmask = __ballot(pass);
            const uint32_t npass = (uint32_t)__popcll(pmask);
            const uint32_t kq = report_psms < npass ? report_psms : npass;  // scoring.rs:284
            __syncthreads();
            if (pass) {
                uint32_t above = 0;
                uint64_t m = pmask;
                while (m) {
                    const uint32_t j = (uint32_t)__ffsll((long long)m) - 1;
                    m &= m - 1;
                    above += j != lane && quick_gt(qk[j], mk);
                }
                if (above < kq) la.keep(keep)[pep] = 1;
            }
            return true;
        }
        pc.mark(2);  // (... ln, hyperscore)
        // stable sort, descending by hyperscore.total_cmp (scoring.rs:495), as a rank computation
        const long long key = order_key64(h);
        const uint64_t pmask = __ballot(pass);
        const uint32_t npass = (uint32_t)__popcll(pmask);
        uint32_t rank = 0;
        double next_h = 0.0, best_h = 0.0;  // hyperscore of the next rank (0 if none) and of rank 0
        bool tie = false;                   // equal hyperscores meet at a reported rank
        if (per_round == 1) {
            // one reported PSM: the sort reduces to the largest key (first lane on ties — the sort is stable) and the
            // largest key among the others
            const long long lowest = (long long)0x8000000000000000ull;
            const long long best_key = wave_max_i64(pass ? key : lowest);
            const uint64_t wins = __ballot(pass && key == best_key);
            const uint32_t wl = (uint32_t)__ffsll((long long)wins) - 1;
            const long long second_key = wave_max_i64(pass && lane != wl ? key : lowest);
            rank = pass && lane == wl ? 0u : 1u;
            best_h = from_order_key64(best_key);
            next_h = npass > 1 ? from_order_key64(second_key) : 0.0;
            tie = (wins & (wins - 1)) != 0ull;
        } else {
            R.s_key[lane] = key;
            __syncthreads();
            if (pass) {
                uint64_t m = pmask;
                while (m) {
                    const uint32_t j = (uint32_t)__ffsll((long long)m) - 1;
                    m &= m - 1;
                    const long long kj = R.s_key[j];
                    rank += (kj > key) || (kj == key && j < lane);
                }
                R.s_sorted[rank] = h;
            }
            __syncthreads();
            if (pass && rank < per_round) {
                next_h = rank + 1 < npass ? R.s_sorted[rank + 1] : 0.0;
                best_h = R.s_sorted[0];
                tie = rank + 1 < npass && __double_as_longlong(R.s_sorted[rank]) == __double_as_longlong(R.s_sorted[rank + 1]);
            }
        }
        if (__builtin_expect(!list_is_exact && __ballot(tie) != 0ull, 0)) {  // (cold: spill code belongs in here, not around it)
            // The preliminary list came from order-free trims, so the stable sort above is only trustworthy when no two equal
            // hyperscores meet at a reported rank (i, i + 1 with i < per_round).  Otherwise the exact heap layout decides.
            // Cheap when ONE PSM is reported: whichever of the tied candidates wins, its Feature is what its lane would report
            // anyway (rank 1, delta_next = delta_best = 0: the runner-up has the same hyperscore) — only WHICH of them comes first in
            // the reference's list has to be found out.  Everything else (several reported PSMs, chimera rounds, several queries or a
            // large window behind the list) takes the exact retry pass.
            // Settled RIGHT HERE when the first pass kept the window counts of this spectrum's (single) query: the wavefront
            // replays bounded_min_heapify from them (Heap32, as the exact path of prelim_spectrum does; the counts staged in the
            // LDS of the bitmap and the peak table, which are dead by now), looks the tied candidates up in the replayed list
            // and the earliest one reports through the ordinary path below — no parking, no extra launch behind the step's last
            // rescoring wavefront.  ~25 us for the 3 % of the wavefronts that get here.
            la.refresh();
            if (queue_on_tie && la.cnt_store() && per_round == 1 && !chimera) {
                const uint32_t* __restrict__ row = la.cnt_store() + (size_t)xcd_position(blockIdx.x, la.batch_n(), la.xcd_chunk()) * la.cnt_stride();
                const uint32_t left = uni(row[0]), potential = uni(row[1]);
                if (potential != 0u && (potential + 1) / 2 <= FAST_TIE_WORDS) {  // (a row that would not fit: the retry pass below)
                    const bool is_best = pass && __double_as_longlong(h) == __double_as_longlong(best_h);
                    uint64_t tb = __ballot(is_best);
                    const uint32_t k = trim_k(potential, report_psms);
                    Counters cnt;
                    cnt.p = pbm;  // [(potential + 1) / 2] words <= FAST_TIE_WORDS: the bitmap + the peak table behind it (carve_rescore)
                    __syncthreads();
                    for (uint32_t i = lane; i < (potential + 1) / 2; i += WAVE) cnt.p[i] = row[CNT_ROW_HEADER + i];
                    __syncthreads();
                    uint32_t hl = 0;  // lane j: entry j of the trimmed list (key `count << 16 | slot`, 0: PreScore::default())
                    if (potential > k) {
                        Heap32 hp;
                        {
                            const uint32_t c = lane < k ? cnt.get(lane) : 0;
                            wh32_init(hp, c ? (c << 16) | lane : 0u, k);
                        }
                        wh32_build(hp, k);
#ifndef SAGE_TIE_REPLAY_PAR
#define SAGE_TIE_REPLAY_PAR 1  // the offers' root replacement by all lanes at once (wh32_replace_root with its per-lane path masks)
#endif
                        const HeapPath path = heap_path_of_lane();
                        for (uint32_t base = k; base < potential; base += WAVE) {
                            const uint32_t i = base + lane;
                            const uint32_t c = i < potential ? cnt.get(i) : 0;
                            const uint32_t v = (c << 16) | i;
                            uint64_t mask = __ballot(c > 0 && c >= (wh32_get(hp, 0) >> 16));
                            while (mask) {
                                const uint32_t bit = (uint32_t)__ffsll((long long)mask) - 1;
                                mask &= mask - 1;
                                const uint32_t vv = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)__builtin_amdgcn_readfirstlane(bit));
                                if (SAGE_TIE_REPLAY_PAR) wh32_offer_par(hp, k, vv, path);
                                else wh32_offer(hp, k, vv);
                            }
                        }
                        hl = hp.h;
                    } else {  // heap.rs:8-10: the slice stays as it is — slot order
                        const uint32_t c = lane < potential ? cnt.get(lane) : 0;
                        hl = c ? (c << 16) | lane : 0u;
                    }
                    uint32_t win = 0, best_pos = 0xFFFFFFFFu;
                    while (tb) {
                        const uint32_t j = (uint32_t)__ffsll((long long)tb) - 1;
                        tb &= tb - 1;
                        const uint32_t slot = (uint32_t)__builtin_amdgcn_readlane((int)pep, (int)j) - left;
                        const uint64_t at = __ballot(lane < k && hl != 0u && (hl & 0xFFFFu) == slot);
                        const uint32_t pos = at ? (uint32_t)__ffsll((long long)at) - 1 : 0xFFFFFFFEu;
                        if (pos < best_pos) { best_pos = pos; win = j; }
                    }
                    rank = pass && lane == win ? 0u : 1u;  // (next_h == best_h already: the runner-up has the same hyperscore)
                    tie = false;
                    if (lane == 0) atomicAdd(la.counters() + CTR_TIE_STRIPE0 + CTR_TIE_STRIDE * (blockIdx.x % CTR_TIE_STRIPES), 1u);  // (statistics: SageTiming::n_tied)
                }
            }
            if (__ballot(tie) != 0ull) {
                if (queue_on_tie && lane == 0) {
                    la.status()[spec] = ST_RETRY;
                    la.retry()[atomicAdd(la.counters() + CTR_RETRY, 1u)] = spec;
                    la.out_count()[spec] = 0;
                }
                return false;
            }
        }
        pc.mark(3);
        la.refresh();
        if (pass && rank < per_round) {  // scoring.rs:504-594
            const uint2 meta = R.meta[lane];
            const uint4 h0 = *(const uint4*)R.hdr;         // tic, precursor mass / charge, rt, ims
            const uint4 h1 = *(const uint4*)(R.hdr + 4);   // file, matched peaks, scored candidates
            const SageFeature f = make_feature(meta.x, __uint_as_float(meta.y), la.spec_base() + spec, pep, z, iso, s, h, next_h, best_h,
                                               chimera ? round + 1 : rank + 1, lambda, ln_lambda, __uint_as_float(h0.y), __uint_as_float(h0.z),
                                               __uint_as_float(h0.w), h1.x, __uint_as_float(h0.x), h1.z, la.lnfact_table(), la.lnfact_n());
            *(SageFeature*)(R.stage + (size_t)(chimera ? 0u : rank) * FEATURE_WORDS) = f;
        }
        const uint32_t emitted = npass < per_round ? npass : per_round;
        // the records of this round leave together: consecutive ranks are consecutive records, so the wavefront stores them as
        // one contiguous run of dwords (full write requests, whether `out` is HBM or the caller's page-locked host memory)
        __syncthreads();
        {
            uint32_t* __restrict__ dst = (uint32_t*)(la.out() + (size_t)spec * report_psms + (chimera ? round : 0u));
            for (uint32_t i = lane; i < emitted * FEATURE_WORDS; i += WAVE) dst[i] = R.stage[i];
        }
        pc.mark(4);
        n_emitted += emitted;
        if (!chimera || emitted == 0 || round + 1 == report_psms) break;  // (chimera: report_psms rounds of one PSM)

        // ---- remove_matched_peaks(winner), scoring.rs:598-644 ----
        const uint64_t wmask = __ballot(pass && rank == 0);
        const uint32_t wl = (uint32_t)__ffsll((long long)wmask) - 1;
        const uint32_t wmfc = __shfl(nfz + 1, wl, 64);
        const uint32_t w_items = __shfl(SAGE_N_ITEMS, wl, 64);
        const unsigned long long w_base = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(ion_base >> 32), (int)wl, 64) << 32) |
                                          (uint32_t)__shfl((int)(uint32_t)ion_base, (int)wl, 64);
        const float* wions = la.ions() + w_base;
        float tic = __uint_as_float(R.hdr[HDR_TIC]);
        remove_matched_peaks_dev(pm, pi, R.rm, R.rm2, P, tic, wions, w_items, wmfc, sc.fragment_tol);
        __syncthreads();
        if (lane == 0) R.hdr[HDR_TIC] = __float_as_uint(tic);
        __syncthreads();
    }
    {
        LateArgs<KA> la(db, sc, b, w, lnfact_table, lnfact_n, out, out_count);
        la.refresh();
        if (lane == 0) la.out_count()[spec] = n_emitted;
    }
    return true;
#undef SAGE_N_ITEMS
}

// ---- rescoring of a preliminary list longer than a wavefront: report_psms > 32, up to BIG_K = 1024 candidates ------------------------
// The same steps as rescore_spectrum — score_candidates / hyperscore / stable sort by hyperscore / Feature records, the chimera
// loop, quick_score's k-select — 64 candidates at a time, the per-candidate results parked in LDS between the steps.  The list is
// always the reference's (exact trims: no tie can be mis-ranked, no retry).  Records leave lane by lane.
struct BigScore {  // what a Feature needs of a candidate's Score
    uint32_t matched_b, matched_y;
    float summed_b, summed_y, ppm_difference;
    uint32_t longest_b, longest_y;
};
__host__ __device__ inline size_t rescore_big_bytes(bool quick, uint32_t kstride) {
    // total_cmp keys of the hyperscores by list position (lowest: did not pass), hyperscores by rank, the scores; quick_score's keys
    // — per entry of the preliminary list (kstride: its length rounded up to a wavefront)
    static_assert(sizeof(BigScore) % 4 == 0 && sizeof(QuickKey) % 8 == 0, "array alignment");
    return (size_t)kstride * (8 + 8 + sizeof(BigScore)) + (quick ? (size_t)kstride * sizeof(QuickKey) : 0) + 8;
}
template <bool LONG, bool HUGE>
__global__ __launch_bounds__(64) void rescore_big_kernel(DevDbView db, DevScorer sc, DevBatchView b, DevWork w,
                                                         const double* __restrict__ lnfact_table, uint32_t lnfact_n,
                                                         SageFeature* __restrict__ out, uint32_t* __restrict__ out_count,
                                                         uint8_t* __restrict__ keep) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t lane = lane_id();
    const size_t scratch = (rescore_scratch_bytes(false) + 15) & ~(size_t)15;
    const size_t fixed = rescore_fixed_bytes(sc, b);
    for (uint32_t blk = blockIdx.x; blk < b.n; blk += gridDim.x) {
        const uint32_t spec = b.order ? b.order[blk] : blk;
        const uint32_t st = w.status[spec];
        __syncthreads();
        if (st != ST_OK && st != ST_OK_ORDERED) {
            if (lane == 0 && !keep) out_count[spec] = 0;
            continue;
        }
        const RescoreLds R = carve_rescore(smem, smem + scratch, b);
        const uint32_t ks = w.kstride;  // (a multiple of 64)
        unsigned char* gp = HUGE ? w.hugebuf + (size_t)blockIdx.x * w.huge_stride : smem + scratch + fixed;
        long long* const g_key = (long long*)gp;
        double* const g_sorted = (double*)(g_key + ks);
        BigScore* const g_score = (BigScore*)(g_sorted + ks);
        QuickKey* const g_qk = (QuickKey*)(((uintptr_t)(g_score + ks) + 7) & ~(uintptr_t)7);
        uint32_t* const pbm = R.pbm;
        uint32_t* const plut = R.plut;
        float* const pm = R.pm;
        float* const pi = R.pi;
        const uint64_t p0 = b.peak_off[spec];
        uint32_t P = (uint32_t)(b.peak_off[spec + 1] - p0);
        for (uint32_t i = lane; i < P; i += WAVE) {
            pm[i] = b.masses[p0 + i];
            pi[i] = b.intensities[p0 + i];
        }
        const uint32_t ncand = w.cand_len[spec] < ks ? w.cand_len[spec] : ks;
        const uint64_t* __restrict__ list = w.cand + (size_t)spec * sc.kmax;
        const uint32_t tot_matched = w.totals[2 * spec], tot_scored = w.totals[2 * spec + 1];
        float tic = b.tic[spec];
        const double lambda = (double)tot_matched / (double)tot_scored;  // scoring.rs:499
        double ln_lambda = 0.0;
        const float mzp = b.precursor_mz[spec] - PROTON;                // scoring.rs:502
        const float rt = b.rt ? b.rt[spec] : 0.0f;
        float ims = 0.0f;
        if (b.ims) { const float v = b.ims[spec]; ims = v == v ? v : 0.0f; }
        const uint32_t fid = b.file_id ? b.file_id[spec] : 0;
        const uint32_t rounds = sc.chimera ? sc.report_psms : 1;
        const uint32_t per_round = sc.chimera ? 1 : sc.report_psms;
        const bool sym_tol = (sc.tol_mode & TOL_SYM) != 0u;
        uint32_t nterm_mask = 0;
        for (uint32_t k = 0; k < db.n_kinds; k++) nterm_mask |= (db.ion_kinds[k] <= 2 ? 1u : 0u) << k;
        const long long lowest = (long long)0x8000000000000000ull;
        uint32_t n_emitted = 0;
        __syncthreads();
        for (uint32_t round = 0; round < rounds; round++) {
            float inv_w;
            build_peak_lut(plut, inv_w, pm, P);
            if (round == 0) build_peak_bitmap(pbm, pm, P, sc.pbm_reach);
            __syncthreads();
            uint32_t npass = 0;
            for (uint32_t base = 0; base < ncand; base += WAVE) {
                const uint32_t i = base + lane;
                const uint64_t mine = i < ncand ? list[i] : PRESCORE_EMPTY;
                const uint32_t pep = prescore_peptide(mine);
                const bool valid = pep != 0xFFFFFFFFu;  // scoring.rs:489
                const uint32_t z = prescore_charge(mine);
                const uint32_t nfz = max_fragment_charge(sc.max_fragment_charge, z) - 1;
                uint64_t ion_base = 0;
                uint32_t lm1 = 0;
                if (valid) {
                    ion_base = db.ion_off[pep];
                    const uint32_t plen = db.pep_info[pep] & 0xFFFFu;
                    lm1 = db.n_kinds && plen ? plen - 1u : 0u;
                }
                const bool any_fz2 = __ballot(valid && nfz >= 2) != 0ull, any_fz3 = __ballot(valid && nfz >= 3) != 0ull;
                Score s;
                s.peptide = 0;
                s.precursor_charge = 0;
                s.isotope_error = 0;
                s.matched_b = s.matched_y = 0;
                s.summed_b = s.summed_y = 0.0f;
                s.ppm_difference = 0.0f;
                s.longest_b = s.longest_y = 0;
                NoClock nc;
                score_candidates<NoClock, LONG>(db, sc, pbm, plut, pm, pi, P, inv_w, valid, ion_base, lm1, nfz, any_fz2, any_fz3, nterm_mask, sym_tol, s, nc);
                double h = 0.0;
                bool pass = false;
                bool ln_undecided;  // (never: both phases)
                const double ln_i = cr_log_pair<true>(hyperscore_arg(s), lambda, round == 0 && base == 0, false, ln_lambda, ln_undecided);
                if (valid) {
                    s.ppm_difference /= s.summed_b + s.summed_y;  // scoring.rs:759
                    h = hyperscore_dev(sc.score_type, s, ln_i, lnfact_table, lnfact_n);
                    pass = (s.matched_b + s.matched_y) >= sc.min_matched_peaks;  // scoring.rs:491
                }
                npass += (uint32_t)__popcll(__ballot(pass));
                if (i < ncand) {
                    g_key[i] = pass ? order_key64(h) : lowest;
                    BigScore q;
                    q.matched_b = s.matched_b; q.matched_y = s.matched_y; q.summed_b = s.summed_b; q.summed_y = s.summed_y;
                    q.ppm_difference = s.ppm_difference; q.longest_b = s.longest_b; q.longest_y = s.longest_y;
                    g_score[i] = q;
                    if (keep) {
                        QuickKey mk;
                        mk.peptide = pep;
                        mk.matched_b = s.matched_b; mk.matched_y = s.matched_y; mk.summed_b = s.summed_b; mk.summed_y = s.summed_y;
                        mk.longest_b = s.longest_b; mk.longest_y = s.longest_y; mk.hyperscore = h; mk.ppm_difference = s.ppm_difference;
                        mk.charge = z; mk.iso = prescore_iso(mine);
                        g_qk[i] = mk;
                    }
                }
            }
            __syncthreads();
            if (keep) {
                // quick_score, prefilter_low_memory (scoring.rs:270-289): the report_psms largest passing candidates by Score's
                // derived order (see rescore_spectrum)
                const uint32_t kq = sc.report_psms < npass ? sc.report_psms : npass;  // scoring.rs:284
                for (uint32_t i = lane; i < ncand; i += WAVE) {
                    const QuickKey mk = g_qk[i];
                    if (g_key[i] == lowest) continue;  // did not pass min_matched_peaks
                    uint32_t above = 0;
                    for (uint32_t j = 0; j < ncand; j++) above += j != i && g_key[j] != lowest && quick_gt(g_qk[j], mk);
                    if (above < kq) keep[mk.peptide] = 1;
                }
                break;
            }
            // stable sort, descending by hyperscore.total_cmp (scoring.rs:495), as a rank computation over the whole list
            for (uint32_t i = lane; i < ncand; i += WAVE) {
                const long long key = g_key[i];
                if (key == lowest) continue;
                uint32_t rank = 0;
                for (uint32_t j = 0; j < ncand; j++) {
                    const long long kj = g_key[j];  // (an LDS broadcast)
                    rank += kj != lowest && ((kj > key) || (kj == key && j < i));
                }
                g_sorted[rank] = from_order_key64(key);
            }
            __syncthreads();
            const uint32_t emitted = npass < per_round ? npass : per_round;
            uint32_t winner = 0xFFFFFFFFu;
            for (uint32_t base = 0; base < ncand; base += WAVE) {
                const uint32_t i = base + lane;
                bool reports = false;
                uint32_t rank = 0;
                long long key = lowest;
                if (i < ncand) key = g_key[i];
                if (key != lowest) {
                    for (uint32_t j = 0; j < ncand; j++) {
                        const long long kj = g_key[j];
                        rank += kj != lowest && ((kj > key) || (kj == key && j < i));
                    }
                    reports = rank < per_round;
                }
                if (reports) {  // scoring.rs:504-594
                    const uint64_t mine = list[i];
                    const BigScore q = g_score[i];
                    Score s;
                    s.peptide = 0; s.precursor_charge = 0; s.isotope_error = 0;
                    s.matched_b = q.matched_b; s.matched_y = q.matched_y; s.summed_b = q.summed_b; s.summed_y = q.summed_y;
                    s.ppm_difference = q.ppm_difference; s.longest_b = q.longest_b; s.longest_y = q.longest_y;
                    const double h = g_sorted[rank];
                    const double next = rank + 1 < npass ? g_sorted[rank + 1] : 0.0, best = g_sorted[0];
                    const SageFeature f = make_feature(db.pep_info[prescore_peptide(mine)], db.pep_mono[prescore_peptide(mine)], b.spec_base + spec,
                                                       prescore_peptide(mine), prescore_charge(mine), prescore_iso(mine), s, h, next, best,
                                                       sc.chimera ? round + 1 : rank + 1, lambda, ln_lambda, mzp, rt, ims, fid, tic, tot_scored, lnfact_table,
                                                       lnfact_n);
                    out[(size_t)spec * sc.report_psms + (sc.chimera ? round : rank)] = f;
                }
                const uint64_t wm = __ballot(reports && rank == 0);
                if (wm) winner = base + (uint32_t)__ffsll((long long)wm) - 1;
            }
            n_emitted += emitted;
            if (!sc.chimera || emitted == 0 || round + 1 == rounds) break;
            // ---- remove_matched_peaks(winner), scoring.rs:598-644 ----
            const uint64_t wmine = list[winner];
            const uint32_t wpep = prescore_peptide(wmine);
            const uint32_t wmfc = max_fragment_charge(sc.max_fragment_charge, prescore_charge(wmine));
            const uint32_t wplen = db.pep_info[wpep] & 0xFFFFu;
            const uint32_t wlm1 = db.n_kinds && wplen ? wplen - 1u : 0u;
            __syncthreads();
            remove_matched_peaks_dev(pm, pi, R.rm, R.rm2, P, tic, db.ions + db.ion_off[wpep], db.n_kinds * wlm1 * (wmfc - 1), wmfc, sc.fragment_tol);
        }
        if (lane == 0 && !keep) out_count[spec] = n_emitted;
    }
}

// scratch bytes of the fused kernel: the preliminary phase's LDS and the rescoring scratch take turns in them
__host__ __device__ inline size_t narrow_scratch_bytes(const DevScorer& sc, const DevBatchView& b) {
    const size_t a = prelim_layout_bytes(sc, b), c = (rescore_scratch_bytes(false) + 15) & ~(size_t)15;
    return a > c ? a : c;
}

// Rescoring as a kernel of its own, behind prelim_kernel / the large-window kernels (candidate lists in HBM).  A spectrum whose
// reported ranks tie in hyperscore is queued for the exact retry pass.  (Settling the tie here — the preliminary phase once more
// with exact trims, inline or behind a call — was measured: the extra code costs the hot path its registers, rescoring went
// from 3.7 to 6.0 resp. 7.2 ms per 500 000 C3 spectra.  DESIGN.md 4.7.)
// FAST: the instance with the short divisions (core.h: div_const_fast), for scorers whose dividends the host has bounded.
template <bool PROF, bool ACC, bool FAST = false>
__global__ __launch_bounds__(64) SAGE_RESCORE_WAVES_ATTR void rescore_kernel(RescoreKernargs A) {
    // (ONE argument: rescore_spectrum reads what it needs late straight from the kernarg segment — LateArgs<RescoreKernargs>)
    const DevDbView& db = A.db;
    const DevScorer& sc = A.sc;
    const DevBatchView& b = A.b;
    const DevWork& w = A.w;
    const double* __restrict__ lnfact_table = A.lnfact_table;
    const uint32_t lnfact_n = A.lnfact_n;
    SageFeature* __restrict__ out = A.out;
    uint32_t* __restrict__ out_count = A.out_count;
    uint8_t* __restrict__ keep = A.keep;
    typedef typename std::conditional<PROF, PhaseClock, NoClock>::type Clock;
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t lane = lane_id();
    if (blockIdx.x >= b.n) return;
    staggered_start(blockIdx.x);
    uint32_t n_batch = b.n;
    if (b.n_dev) {  // retry pass: device-side count
        n_batch = *b.n_dev < n_batch ? *b.n_dev : n_batch;
        if (blockIdx.x >= n_batch) return;
    }
    const uint32_t pos = xcd_position(blockIdx.x, n_batch, sc.xcd_chunk);
    // (with schedule records — DevBatchView::sched — the block's spectrum and its peak range come in one trip, and the peaks are
    // requested beside the status and the list instead of behind peak_off[spec])
    uint4 rec = make_uint4(0u, 0u, 0u, 0u);
    if (b.sched) rec = b.sched[2 * (size_t)pos];
    const uint32_t spec = b.sched ? uni(rec.x) : b.order ? b.order[pos] : pos;
    // Everything that hangs on `spec` alone is requested together, ahead of the first use: the spectrum's status, the whole row
    // of its preliminary list (unconditionally — the array is padded by a wavefront — so that the load does not wait for the
    // list's length), the length, the totals, the peak range.  The chain of dependent round trips is then
    // schedule record -> {status, list, peaks} -> ion offsets -> ions (without the records: order -> {status, list, peak range} ->
    // {peaks, ion offsets} -> ions).
    const uint32_t st = w.status[spec];
    const uint64_t row_word = w.cand[(size_t)spec * sc.kmax + lane];
    const uint32_t ncand = w.cand_len[spec];
    const uint32_t tot_m = w.totals[2 * spec], tot_s = w.totals[2 * spec + 1];
    const uint64_t p0 = b.sched ? ((uint64_t)uni(rec.w) << 32) | uni(rec.z) : b.peak_off[spec];
    const uint32_t P = b.sched ? uni(rec.y) : (uint32_t)(b.peak_off[spec + 1] - p0);
#ifndef SAGE_EARLY_PEAKS
#define SAGE_EARLY_PEAKS 1  // with schedule records: the first 192 peaks requested HERE, in front of the branches on the status
#endif
    // (the compiler does not move a load above the early returns below, so the peaks used to wait for the status to arrive)
    float em0 = 0.f, em1 = 0.f, em2 = 0.f, ei0 = 0.f, ei1 = 0.f, ei2 = 0.f;
    const bool early = SAGE_EARLY_PEAKS && b.sched != nullptr;
    if (early) {
        const float* __restrict__ gm = b.masses + p0;
        const float* __restrict__ gi = b.intensities + p0;
        if (lane < P) { em0 = gm[lane]; ei0 = gi[lane]; }
        if (lane + WAVE < P) { em1 = gm[lane + WAVE]; ei1 = gi[lane + WAVE]; }
        if (lane + 2 * WAVE < P) { em2 = gm[lane + 2 * WAVE]; ei2 = gi[lane + 2 * WAVE]; }
    }
    if (st == ST_DONE) return;  // reported by the fused narrow kernel of this pass
    if (st != ST_OK && st != ST_OK_ORDERED) {
        if (lane == 0 && !keep) out_count[spec] = 0;
        return;
    }
    const RescoreLds R = carve_rescore(smem, smem + ((rescore_scratch_bytes(keep != nullptr) + 15) & ~(size_t)15), b);
    Clock pc;
    pc.start((sc.dbg_flags & 512u) && !sc.exact ? nullptr : w.dbg, blockIdx.x, 1);
    if (early) {
        if (lane < P) { R.pm[lane] = em0; R.pi[lane] = ei0; }
        if (lane + WAVE < P) { R.pm[lane + WAVE] = em1; R.pi[lane + WAVE] = ei1; }
        if (lane + 2 * WAVE < P) { R.pm[lane + 2 * WAVE] = em2; R.pi[lane + 2 * WAVE] = ei2; }
    }
    for (uint32_t i = early ? lane + 3 * WAVE : lane; i < P; i += WAVE) {
        R.pm[i] = b.masses[p0 + i];
        R.pi[i] = b.intensities[p0 + i];
    }
    const uint64_t mine = lane < ncand ? row_word : PRESCORE_EMPTY;
    // (a list no trim touched is the reference's list already: equal hyperscores are ranked by it, no retry)
    rescore_spectrum<ACC, RescoreKernargs, FAST>(db, sc, b, w, lnfact_table, lnfact_n, out, out_count, keep, R, spec, P, mine, tot_m, tot_s,
                                                 sc.exact != 0 || st == ST_OK_ORDERED, true, pc);
}

#ifdef SAGE_HIP_EXPERIMENTS  // (measured slower than the two kernels, DESIGN.md 4.7: compiled only into experiment builds)
// ---- the first pass of a narrow search as ONE launch of two kinds of workgroups ----------------------------------------------------
// Matching + k-select (prelim_spectrum) and rescoring (rescore_spectrum) are separate wavefronts, as in prelim_kernel /
// rescore_kernel — each body keeps its own register allocation; one wavefront doing both (narrow_kernel) pays 70-100 spills — but
// they share a launch: workgroup ids alternate between the two roles in groups of eight (one of each kind per XCD), the
// rescoring workgroup of a spectrum trailing its preliminary workgroup by SEARCH_LAG ids, and the hand-over is a per-spectrum
// word in HBM (DevWork::ready == the launch's epoch; payload and word are agent-scope write-through stores, the consumer polls the
// word and reads the payload with agent-scope loads — no fences: a release per spectrum writes the L2 back every time).
// What that buys over two launches: one cold start instead of two (every launch begins with empty L2s: ~0.13 ms per kernel on
// C3, a third of a 62 500-spectrum step), no drain between the phases, and CUs that run memory-bound matching next to VALU-bound
// rescoring all the time.  Forward progress: a workgroup only ever waits for one with a smaller id, which was dispatched before
// it and waits for nobody.
constexpr uint32_t SEARCH_LAG = 8192;  // ids (a multiple of 8): ~1.6 x the wavefronts resident on the GPU, so the wait is rare
struct SearchRole {
    bool rescoring;
    uint32_t v;  // the role's virtual workgroup id (v % 8 == the XCD it runs on, like a plain launch of that role)
};
__host__ __device__ inline uint32_t search_slots(uint32_t n) { return (n + 7u) & ~7u; }
__host__ __device__ inline uint32_t search_lag(uint32_t n, uint32_t want) {
    want = want ? (want + 7u) & ~7u : SEARCH_LAG;
    return search_slots(n) < want ? search_slots(n) : want;
}
__host__ __device__ inline SearchRole search_role(uint32_t b, uint32_t n, uint32_t want_lag) {
    const uint32_t np = search_slots(n), lag = search_lag(n, want_lag);
    SearchRole r;
    if (b < lag) {  // head: preliminary only
        r.rescoring = false;
        r.v = b;
    } else if (b < 2 * np - lag) {  // groups of eight, alternating
        const uint32_t q = (b - lag) >> 3, i = (b - lag) & 7u;
        r.rescoring = (q & 1u) != 0;
        r.v = (r.rescoring ? 0u : lag) + (q >> 1) * 8u + i;
    } else {  // tail: the last `lag` rescoring workgroups
        r.rescoring = true;
        r.v = np - lag + (b - (2 * np - lag));
    }
    return r;
}
template <bool PROBE, bool PROF>
__global__ __launch_bounds__(64) SAGE_PRELIM_WAVES_ATTR void search_kernel(DevDbView db, DevScorer sc, DevBatchView b, DevWork w,
                                                                           const double* __restrict__ lnfact_table, uint32_t lnfact_n,
                                                                           SageFeature* __restrict__ out, uint32_t* __restrict__ out_count) {
    typedef typename std::conditional<PROF, PhaseClock, NoClock>::type Clock;
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t lane = lane_id();
    const SearchRole role = search_role(blockIdx.x, b.n, w.search_lag);
    if (role.v >= b.n) return;
    const uint32_t pos = xcd_position(role.v, b.n, sc.xcd_chunk);
    const uint32_t spec = b.order ? uni(b.order[pos]) : pos;
    if (!role.rescoring) {
        const PrelimLds L = carve_prelim(smem, sc, b);
        Clock pc;
        pc.start(w.dbg, role.v, 0);
        const SpecInfo si = load_spec(sc, b, spec);
        const PrelimResult r = prelim_spectrum<PROBE>(db, sc, b, L, si, false, pc);
        // hand-over (cdna_hip_programming.md, guideline 16): the payload goes out WRITE-THROUGH (agent-scope stores: no release fence,
        // which would write back the whole L2 once per spectrum — measured: 4 x the kernel time), the wavefront waits for its
        // stores, one lane sets the word
        if (r.deferred) {
            if (lane == 0) {
                __hip_atomic_store(w.status + spec, (uint32_t)ST_DEFERRED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t it = atomicAdd(w.n_deferred + CTR_QUEUED, 1u);
                w.queue[it] = spec;  // (read by the kernels of later launches)
                w.item_of[spec] = it;
            }
        } else {
            if (lane == 0) {
                if (!r.ok) atomicAdd(w.n_deferred + CTR_LIST_OVERFLOW, 1u);
                __hip_atomic_store(w.status + spec, (uint32_t)(r.ok ? (r.untrimmed ? ST_OK_ORDERED : ST_OK) : ST_OVERFLOW), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(w.cand_len + spec, r.stored, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(w.totals + 2 * spec, r.matched, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(w.totals + 2 * spec + 1, r.scored, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            for (uint32_t i = lane; i < r.stored; i += WAVE)
                __hip_atomic_store(w.cand + (size_t)spec * sc.kmax + i, L.listB[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        pc.mark(4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(w.ready + spec, w.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // ---- rescoring role ----
    const RescoreLds R = carve_rescore(smem, smem + ((rescore_scratch_bytes(false) + 15) & ~(size_t)15), b);
    Clock pc;
    pc.start(w.dbg, role.v, 1);
    const uint64_t p0 = b.peak_off[spec];
    const uint32_t P = (uint32_t)(b.peak_off[spec + 1] - p0);
    for (uint32_t i = lane; i < P; i += WAVE) {  // (the peaks do not depend on the preliminary workgroup: in flight under the wait)
        R.pm[i] = b.masses[p0 + i];
        R.pi[i] = b.intensities[p0 + i];
    }
    // the consumer polls the one word (relaxed), then reads the payload with agent-scope loads: they take the vector path around
    // this XCD's L2 lines and the scalar cache, which may hold the previous step's values of the same addresses
    while (__hip_atomic_load(w.ready + spec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != w.epoch) __builtin_amdgcn_s_sleep(32);
    asm volatile("" ::: "memory");
    const uint32_t st = __hip_atomic_load(w.status + spec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (st != ST_OK && st != ST_OK_ORDERED) {  // queued for the large-window kernels (their rescoring follows in a launch of its own), or overflowed
        if (lane == 0 && st == ST_OVERFLOW) out_count[spec] = 0;
        return;
    }
    const uint32_t ncand = __hip_atomic_load(w.cand_len + spec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t tot_m = __hip_atomic_load(w.totals + 2 * spec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t tot_s = __hip_atomic_load(w.totals + 2 * spec + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t mine = lane < ncand ? __hip_atomic_load(w.cand + (size_t)spec * sc.kmax + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                       : PRESCORE_EMPTY;
    const bool done = rescore_spectrum<true, void>(db, sc, b, w, lnfact_table, lnfact_n, out, out_count, nullptr, R, spec, P, mine, tot_m, tot_s,
                                       st == ST_OK_ORDERED, true, pc);
    if (done && lane == 0) w.status[spec] = ST_DONE;  // (a rescore_kernel behind the large-window kernels leaves it alone)
}

#endif  // SAGE_HIP_EXPERIMENTS

// ---- the narrow search in ONE launch ---------------------------------------------------------------------------------------------
// Scorer::score of a spectrum whose precursor windows fit the LDS counters: preliminary matching + k-select (prelim_spectrum)
// and rescoring (rescore_spectrum) by the same wavefront; the candidate list never leaves the chip.  Spectra with a larger
// window are queued for the large-window kernels exactly as prelim_kernel queues them.
// Two uses.  (1) THE EXACT RETRY PASS (DevScorer::exact, spectrum list + count on the device): the spectra whose reported ranks
// tied in rescore_kernel, again with bounded_min_heapify replayed — one launch, no candidate round trip, so the latency chain
// a small batch pays for its ~3 % tied spectra is one wavefront's lifetime.  (2) SAGE_HIP_FUSED=1: the whole first pass, order-free
// trims first and, on a tie, the same wavefront once more with exact trims.  On MI355X (2) is 10-17 % slower than prelim_kernel +
// rescore_kernel although it executes the same instructions: the union of the two phases leaves the register allocator with
// 70-100 spills to scratch (each a memory round trip the wavefront waits for) where the separate kernels have 0 and ~30
// (profiles/r03_fused_vs_separate.md).
template <bool PROBE, bool PROF>
__global__ __launch_bounds__(64) SAGE_NARROW_WAVES_ATTR void narrow_kernel(RescoreKernargs A) {
    // (rescore_kernel's argument struct, `keep` unused; its first four members are prelim_kernel's: both phases read their
    // arguments from the kernarg segment where they use them — ArgRef<PrelimKernargs>, LateArgs<RescoreKernargs>)
    static_assert(offsetof(RescoreKernargs, db) == offsetof(PrelimKernargs, db) && offsetof(RescoreKernargs, sc) == offsetof(PrelimKernargs, sc) &&
                      offsetof(RescoreKernargs, b) == offsetof(PrelimKernargs, b) && offsetof(RescoreKernargs, w) == offsetof(PrelimKernargs, w),
                  "the preliminary phase reads the segment as a PrelimKernargs");
    const DevDbView& db = A.db;
    const DevScorer& sc = A.sc;
    const DevBatchView& b = A.b;
    const DevWork& w = A.w;
    const double* __restrict__ lnfact_table = A.lnfact_table;
    const uint32_t lnfact_n = A.lnfact_n;
    SageFeature* __restrict__ out = A.out;
    uint32_t* __restrict__ out_count = A.out_count;
    typedef typename std::conditional<PROF, PhaseClock, NoClock>::type Clock;
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t lane = lane_id();
    uint32_t n_batch = b.n;
    if (b.n_dev) {  // retry pass: the count is a device-side counter of the first pass
        const uint32_t nd = uni(*b.n_dev);
        n_batch = nd < n_batch ? nd : n_batch;
    }
    const PrelimLds L = carve_prelim(smem, sc, b);
    const RescoreLds R = carve_rescore(smem, smem + narrow_scratch_bytes(sc, b), b);
#pragma unroll 1
    for (uint32_t blk = blockIdx.x; blk < n_batch; blk += gridDim.x) {
        const uint32_t pos = xcd_position(blk, n_batch, sc.xcd_chunk);
        const uint32_t spec = b.order ? uni(b.order[pos]) : pos;
        const SpecInfo si = load_spec(sc, b, spec);
        bool exact = sc.exact != 0;
        uint32_t final_status = ST_DONE;
#pragma unroll 1
        for (;;) {
            __syncthreads();
            Clock pc;
            pc.start((sc.dbg_flags & 512u) && !sc.exact ? nullptr : w.dbg, blk, 0);  // (SAGE_HIP_DEBUG_FLAGS=512: clocks of the exact retry pass only)
            // the rescoring phase's copy of the peaks (its own: the chimera loop edits it); in flight under the window query
            for (uint32_t i = lane; i < si.P; i += WAVE) {
                R.pm[i] = b.masses[si.p0 + i];
                R.pi[i] = b.intensities[si.p0 + i];
            }
            const PrelimResult r = prelim_spectrum<PROBE, false, PrelimKernargs>(db, sc, b, L, si, exact, pc);
            if (r.deferred) {
                if (lane == 0) {
                    const uint32_t it = atomicAdd(w.n_deferred + CTR_QUEUED, 1u);
                    w.queue[it] = spec;
                    if (!w.reuse) w.item_of[spec] = it;  // (the retry pass finds the first pass's records of this spectrum through it)
                }
                final_status = ST_DEFERRED;
                break;
            }
            if (!r.ok) {
                if (lane == 0) {
                    atomicAdd(w.n_deferred + CTR_LIST_OVERFLOW, 1u);
                    out_count[spec] = 0;
                }
                final_status = ST_OVERFLOW;
                break;
            }
            const uint64_t mine = lane < r.stored ? L.listB[lane] : PRESCORE_EMPTY;
            pc.mark(4);
            pc.rebase(1);  // (the rescoring phase accounts under kernel 1)
            __syncthreads();  // the list is in registers: the preliminary phase's LDS is free
            if (rescore_spectrum<true, RescoreKernargs>(db, sc, b, w, lnfact_table, lnfact_n, out, out_count, nullptr, R, spec, si.P, mine, r.matched,
                                                        r.scored, exact || r.untrimmed, false, pc) ||
                exact)
                break;
            exact = true;  // equal hyperscores at a reported rank: once more, with bounded_min_heapify replayed (heap.rs:7-28)
            if (lane == 0) atomicAdd(w.n_deferred + CTR_TIED, 1u);
        }
        if (lane == 0) w.status[spec] = final_status;
    }
}

// The way home of a step's small results in ONE launch (page-locked, mapped destinations): the PSM counts of every spectrum and
// the counter blocks of the step's parts.  Three copy commands at the end of a 1 ms step cost ~60 us of command gaps.
// The counter blocks are left ZEROED for the next step (its first command is then a kernel, not a fill).
// `order` (may be null): the spectra this launch answers for — a part of a step sends the counts of ITS spectra (a range of the
// launch schedule, scattered over the count array), so that no part has to wait for another one's kernels.
__global__ __launch_bounds__(256) void epilogue_kernel(const uint32_t* __restrict__ counts, uint32_t n, uint32_t* __restrict__ h_counts,
                                                       const uint32_t* __restrict__ order, EpilogueParts parts) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t spec = order ? order[i] : i;
        h_counts[spec] = counts[spec];
    }
    if (blockIdx.x == 0)
        for (uint32_t p = 0; p < parts.n; p++)
            for (uint32_t i = threadIdx.x; i < 2 * CTR_COUNT; i += blockDim.x) {
                parts.dst[p][i] = parts.src[p][i];
                parts.src[p][i] = 0u;
            }
}

// quick_score without prefilter_low_memory (scoring.rs:290-296): every peptide of the trimmed preliminary list
__global__ __launch_bounds__(256) void quick_mark_kernel(DevScorer sc, uint32_t n, DevWork w, uint8_t* __restrict__ keep) {
    const uint32_t spec = blockIdx.x * 4 + threadIdx.x / 64, lane = threadIdx.x & 63u;
    if (spec >= n || (w.status[spec] != ST_OK && w.status[spec] != ST_OK_ORDERED)) return;
    for (uint32_t i = lane; i < w.cand_len[spec]; i += WAVE) {
        const uint32_t pep = prescore_peptide(w.cand[(size_t)spec * sc.kmax + i]);
        if (pep != 0xFFFFFFFFu) keep[pep] = 1;
    }
}

// Fragments of the reported PSMs (annotate_matches, scoring.rs:722-752): one wavefront per spectrum replays
// score_candidate's (kind, ion index, fragment charge) loop for every reported PSM and writes the matches in that
// order; with chimera the winner's peaks are removed before the next PSM exactly as score_chimera_fast does.
__global__ __launch_bounds__(64) void annotate_kernel(DevDbView db, DevScorer sc, DevBatchView b, const SageFeature* __restrict__ feats,
                                                      const uint32_t* __restrict__ counts, const uint64_t* __restrict__ psm_off,
                                                      DevFragments out) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t lane = lane_id();
    const uint32_t spec = blockIdx.x;
    if (spec >= b.n) return;
    float* pm = (float*)smem;
    float* pi = pm + b.pcap;
    uint8_t* rm = (uint8_t*)(pi + b.pcap);
    uint8_t* rm2 = rm + b.pcap;
    const uint64_t p0 = b.peak_off[spec];
    uint32_t P = (uint32_t)(b.peak_off[spec + 1] - p0);
    for (uint32_t i = lane; i < P; i += WAVE) {
        pm[i] = b.masses[p0 + i];
        pi[i] = b.intensities[p0 + i];
    }
    __syncthreads();
    float tic = 0.0f;
    const uint32_t count = counts[spec];
    for (uint32_t r = 0; r < count; r++) {
        const size_t slot = (size_t)spec * sc.report_psms + r;
        const SageFeature f = feats[slot];
        const uint32_t pep = f.peptide_idx;
        const uint32_t mfc = max_fragment_charge(sc.max_fragment_charge, f.charge), nfz = mfc - 1;
        const uint64_t ion_base = db.ion_off[pep];
        const uint32_t lm1 = db.n_kinds ? (uint32_t)((db.ion_off[pep + 1] - ion_base) / db.n_kinds) : 0;
        const uint32_t n_items = db.n_kinds * lm1 * nfz;
        const uint32_t ptop = pow2_floor(P);
        uint64_t pos = psm_off[slot];
        for (uint32_t base = 0; base < n_items; base += WAVE) {
            const uint32_t t = base + lane;
            int pk = -1;
            uint32_t ion = 0, charge = 1;
            float mz = 0.0f;
            if (t < n_items) {
                ion = t / nfz;
                charge = t - ion * nfz + 1;
                mz = db.ions[ion_base + ion] / (float)charge;
                pk = select_most_intense_peak_lockstep(pm, pi, P, ptop, mz, sc.fragment_tol);
            }
            const uint64_t m = __ballot(pk >= 0);
            if (pk >= 0) {
                const uint64_t o = pos + (uint64_t)__popcll(m & ((1ull << lane) - 1ull));
                if (o < out.capacity) {
                    const uint32_t kidx = ion / lm1, idx = ion - kidx * lm1;
                    const uint8_t kind = db.ion_kinds[kidx];
                    out.kinds[o] = kind;
                    out.charges[o] = (int32_t)charge;
                    // scoring.rs:739-744: b-like ions count from the N-terminus, y-like from the C-terminus
                    out.fragment_ordinals[o] = kind <= 2 ? (int32_t)idx + 1 : (int32_t)lm1 - (int32_t)idx;
                    out.intensities[o] = pi[pk];
                    out.mz_calculated[o] = mz + PROTON;       // scoring.rs:723
                    out.mz_experimental[o] = pm[pk] + PROTON;  // scoring.rs:722
                }
            }
            pos += (uint64_t)__popcll(m);
        }
        if (sc.chimera && r + 1 < count)
            remove_matched_peaks_dev(pm, pi, rm, rm2, P, tic, db.ions + ion_base, n_items, mfc, sc.fragment_tol);
    }
}

}  // namespace

size_t prelim_lds_bytes(const DevScorer& sc, const DevBatchView& b, bool huge) { return prelim_layout_bytes(sc, b, huge); }
// bytes of DevWork::hugebuf one workgroup of the wide-list kernels needs (the largest of: the preliminary kernel's lists + heap,
// the assembler's lists, the replay's heap, the rescoring kernel's per-candidate arrays), a multiple of 256
size_t huge_stride_bytes(const DevScorer& sc) {
    const uint32_t ks = ((sc.kmax + 63u) / 64u) * 64u;
    size_t n = prelim_big_bytes(sc);
    n = n > assemble_lds_bytes(sc) ? n : assemble_lds_bytes(sc);
    n = n > rescore_big_bytes(true, ks) ? n : rescore_big_bytes(true, ks);
    n = n > (size_t)ks * 8 ? n : (size_t)ks * 8;
    return (n + 255) & ~(size_t)255;
}
size_t tile_lds_bytes(const DevDbView& db, const DevScorer&, const DevBatchView& b, bool cnt8, bool wing) {
    return tile_lds_layout(db.tile_shift, b, nullptr, nullptr, cnt8, wing);
}
// report_psms > 128: lists, heaps and per-candidate scores of up to 1024 entries need more than the 64 KB a kernel gets by default
size_t assemble_lds_bytes(const DevScorer& sc) {
    const bool fold = sc.min_isotope_err != sc.max_isotope_err;
    return ((size_t)sc.list_cap * (fold ? 16 : 8) + 15) & ~(size_t)15;
}
uint32_t fast_tie_lds_words() { return FAST_TIE_WORDS; }
uint32_t huge_grid() { return HUGE_GRID; }
int bigk_kernel_prepare(size_t max_lds_bytes) {
    for (const void* f : {(const void*)prelim_kernel<true, false, true>, (const void*)prelim_kernel<false, false, true>,
                          (const void*)prelim_kernel<true, false, true, true>, (const void*)prelim_kernel<false, false, true, true>,
                          (const void*)rescore_big_kernel<false, false>, (const void*)rescore_big_kernel<true, false>,
                          (const void*)rescore_big_kernel<false, true>, (const void*)rescore_big_kernel<true, true>,
                          (const void*)tile_assemble_kernel<true>, (const void*)tile_assemble_kernel<true, true>,
                          (const void*)tile_replay_big_kernel<false>, (const void*)tile_replay_big_kernel<true>}) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    return (int)hipSuccess;
}
int tile_kernel_prepare(size_t max_lds_bytes) {
    for (const void* f : {(const void*)tile_count_kernel, (const void*)tile_count8_kernel, (const void*)tile_count_wing_kernel}) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    return (int)hipSuccess;
}
// ... and of the per-spectrum kernels: a spectrum of thousands of peaks needs more than the 64 KB a launch gets by default (10 bytes
// of LDS per peak in the rescoring kernels, 4 in the preliminary kernel's probe variant).  Raising the limit changes nothing for the
// launches that stay below it.
int spectrum_kernel_prepare(size_t max_lds_bytes) {
    for (const void* f : {(const void*)prelim_kernel<true, true>, (const void*)prelim_kernel<true, false>, (const void*)prelim_kernel<false, true>,
                          (const void*)prelim_kernel<false, false>, (const void*)rescore_kernel<true, true>, (const void*)rescore_kernel<false, false>,
                          (const void*)rescore_kernel<false, false, true>,
                          (const void*)rescore_kernel<false, true>, (const void*)narrow_kernel<true, true>, (const void*)narrow_kernel<true, false>,
                          (const void*)narrow_kernel<false, true>, (const void*)narrow_kernel<false, false>, (const void*)annotate_kernel}) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds_bytes);
        if (e != hipSuccess) return (int)e;
        // (pbm_bit reads the peak bitmap through LDS address 0: the dynamic LDS of these kernels must start there)
        hipFuncAttributes at;
        if (hipFuncGetAttributes(&at, f) == hipSuccess && at.sharedSizeBytes != 0) return (int)hipErrorInvalidConfiguration;
    }
#ifdef SAGE_HIP_EXPERIMENTS  // (the one-launch experiment's instances take the same limit: ADVICE r05)
    for (const void* f : {(const void*)search_kernel<true, true>, (const void*)search_kernel<true, false>, (const void*)search_kernel<false, true>,
                          (const void*)search_kernel<false, false>}) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
#endif
    return (int)hipSuccess;
}
uint32_t queries_per_spectrum(const DevScorer& sc) {
    const uint32_t n_iso = sc.min_isotope_err != sc.max_isotope_err ? (uint32_t)(sc.max_isotope_err - sc.min_isotope_err) + 1 : 1;
    return (sc.max_precursor_charge - sc.min_precursor_charge + 1) * n_iso;
}
size_t rescore_lds_bytes(const DevScorer& sc, const DevBatchView& b, uint32_t, bool quick, bool huge) {
    if (sc.big_path)  // report_psms > 32 (or long peptides): rescore_big_kernel
        return ((rescore_scratch_bytes(false) + 15) & ~(size_t)15) + rescore_fixed_bytes(sc, b) +
               (huge ? 0 : rescore_big_bytes(quick, ((sc.kmax + 63u) / 64u) * 64u));
    return ((rescore_scratch_bytes(quick) + 15) & ~(size_t)15) + rescore_fixed_bytes(sc, b);
}
size_t narrow_lds_bytes(const DevScorer& sc, const DevBatchView& b) { return narrow_scratch_bytes(sc, b) + rescore_fixed_bytes(sc, b); }

#ifdef SAGE_HIP_EXPERIMENTS
size_t search_lds_bytes(const DevScorer& sc, const DevBatchView& b) {
    const size_t a = prelim_layout_bytes(sc, b), r = rescore_lds_bytes(sc, b, 0, false);
    return a > r ? a : r;
}
void launch_search(const DevDbView& db, const DevScorer& sc, const DevBatchView& b, const DevWork& w, const double* lnfact_table,
                   uint32_t lnfact_n, SageFeature* out, uint32_t* out_count, void* stream) {
    if (b.n == 0) return;
    auto k = b.probe ? (w.dbg ? search_kernel<true, true> : search_kernel<true, false>)
                     : (w.dbg ? search_kernel<false, true> : search_kernel<false, false>);
    hipLaunchKernelGGL(k, dim3(2 * search_slots(b.n)), dim3(64), search_lds_bytes(sc, b), (hipStream_t)stream, db, sc, b, w, lnfact_table,
                       lnfact_n, out, out_count);
}
#else
size_t search_lds_bytes(const DevScorer&, const DevBatchView&) { return 0; }
void launch_search(const DevDbView&, const DevScorer&, const DevBatchView&, const DevWork&, const double*, uint32_t, SageFeature*, uint32_t*, void*) {}
#endif
void launch_narrow(const DevDbView& db, const DevScorer& sc, const DevBatchView& b, const DevWork& w, const double* lnfact_table,
                   uint32_t lnfact_n, SageFeature* out, uint32_t* out_count, void* stream) {
    if (b.n == 0) return;
    auto k = b.probe ? (w.dbg ? narrow_kernel<true, true> : narrow_kernel<true, false>)
                     : (w.dbg ? narrow_kernel<false, true> : narrow_kernel<false, false>);
    // (a retry pass holds a few per cent of the batch: a capped grid strides over the device-side list)
    const uint32_t grid = b.n_dev ? (b.n < RETRY_GRID_CAP ? b.n : RETRY_GRID_CAP) : b.n;
    const RescoreKernargs args{db, sc, b, w, lnfact_table, lnfact_n, out, out_count, nullptr};
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), narrow_lds_bytes(sc, b), (hipStream_t)stream, args);
}
void launch_prelim(const DevDbView& db, const DevScorer& sc, const DevBatchView& b, const DevWork& w, void* stream) {
    if (b.n == 0) return;
    auto k = b.probe ? (w.dbg ? prelim_kernel<true, true> : prelim_kernel<true, false>)
                     : (w.dbg ? prelim_kernel<false, true> : prelim_kernel<false, false>);
    if (sc.big_path) k = b.probe ? prelim_kernel<true, false, true> : prelim_kernel<false, false, true>;  // report_psms > 32
    if (sc.big_path && w.hugebuf) k = b.probe ? prelim_kernel<true, false, true, true> : prelim_kernel<false, false, true, true>;
    // (lists in global memory: a workspace slice per workgroup, so the grid is capped and strides over the batch)
    uint32_t grid = sc.big_path && w.hugebuf && b.n > HUGE_GRID ? HUGE_GRID : b.n;
    // (SAGE_HIP_PRELIM_GRID: workgroups that stride over the batch instead of one per spectrum — an experiment knob, a multiple of 8
    // so that a workgroup's spectra stay on its XCD's share of the schedule)
    static const uint32_t grid_cap = [] { const char* e = getenv("SAGE_HIP_PRELIM_GRID"); return e ? (uint32_t)strtoul(e, nullptr, 10) & ~7u : 0u; }();
    if (grid_cap && grid > grid_cap) grid = grid_cap;
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), prelim_lds_bytes(sc, b, sc.big_path && w.hugebuf), (hipStream_t)stream, PrelimKernargs{db, sc, b, w});
}
void launch_queue(const DevScorer& sc, const DevBatchView& b, const DevWork& w, void* stream) {
    if (b.n == 0) return;
    const uint32_t blocks = (b.n + 255u) / 256u;
    hipLaunchKernelGGL(queue_kernel, dim3(blocks < 4096u ? blocks : 4096u), dim3(256), 0, (hipStream_t)stream, sc, b, w);
}
void launch_prelim_tile(const DevDbView& db, const DevScorer& sc, const DevBatchView& b, const DevWork& w, void* stream,
                        const SideStream* side) {
    if (b.n == 0 || w.tile_blocks == 0) return;
    const bool fold = sc.min_isotope_err != sc.max_isotope_err;
    if (w.winbuf)
        hipLaunchKernelGGL(tile_count_wing_kernel, dim3(w.tile_blocks < b.n ? w.tile_blocks : b.n), dim3(TILE_THREADS),
                           tile_lds_bytes(db, sc, b, false, true), (hipStream_t)stream, TileParams{db, sc, b, w});
    else if (w.cnt8)
        hipLaunchKernelGGL(tile_count8_kernel, dim3(w.tile_blocks < b.n ? w.tile_blocks : b.n), dim3(TILE_THREADS),
                           tile_lds_bytes(db, sc, b, true), (hipStream_t)stream, TileParams{db, sc, b, w});
    else
        hipLaunchKernelGGL(tile_count_kernel, dim3(w.tile_blocks < b.n ? w.tile_blocks : b.n), dim3(TILE_THREADS),
                           tile_lds_bytes(db, sc, b, false), (hipStream_t)stream, TileParams{db, sc, b, w});
    if (hipPeekAtLastError() != hipSuccess) return;  // (never let the kernels below walk records the count kernel did not write)
    const uint64_t nq = (uint64_t)b.n * w.qmax;
    const uint32_t grid_cap = sc.big_path && w.hugebuf ? HUGE_GRID : TILE_GRID_CAP;
    auto capped = [grid_cap](uint64_t blocks) { return (uint32_t)(blocks < grid_cap ? blocks : grid_cap); };
    const size_t assemble_lds = sc.big_path && w.hugebuf ? 0 : assemble_lds_bytes(sc);
    (void)fold;
    if (sc.big_path) {  // report_psms > 32: heaps in LDS (or in the global workspace), always exact
        if (w.hugebuf) {
            hipLaunchKernelGGL(tile_replay_big_kernel<true>, dim3(capped(nq)), dim3(64), WAVE * 4, (hipStream_t)stream, sc, w);
            hipLaunchKernelGGL((tile_assemble_kernel<true, true>), dim3(capped(b.n)), dim3(64), assemble_lds, (hipStream_t)stream, sc, b, w);
        } else {
            hipLaunchKernelGGL(tile_replay_big_kernel<false>, dim3(capped(nq)), dim3(64), (size_t)w.kstride * 8 + WAVE * 4, (hipStream_t)stream, sc, w);
            hipLaunchKernelGGL(tile_assemble_kernel<true>, dim3(capped(b.n)), dim3(64), assemble_lds, (hipStream_t)stream, sc, b, w);
        }
        return;
    }
    // bounded_min_heapify replay: a wavefront per query (the default for every query since the round-6 root replacement —
    // wh32_replace_root: ~12 vector instructions per offer —, which made the lane-per-query kernel's best case, many short streams,
    // no faster: C5's retry pass 6.6 ms whichever way the queries are split below 512 stream words).  The lane-per-query kernel
    // stays for DevWork::replay_split settings that ask for it (tests; SAGE_HIP_REPLAY_WAVE_MAX / _LANE_MAX).
    uint64_t wave_max = w.replay_split;
    if (!sc.exact && (wave_max & 0xFFFFFFFFull)) wave_max |= 0xFFFFFFFFull;
    const bool both = (wave_max & 0xFFFFFFFFull) != 0xFFFFFFFFull && (sc.exact || !(wave_max & 0xFFFFFFFFull));
    // The kernels of this stage take disjoint sets of queries (order-free mode: tile_select_kernel every query whose histogram is
    // whole, the replay the few whose histogram is clipped — a few hundred serial wavefronts that used to run alone on the GPU for
    // 0.7 - 1.1 ms behind the select; the two replay kernels: see replay_by_wavefront): side by side when the caller lends a second
    // stream.
    const bool select = !sc.exact;
    hipStream_t side_stream = (hipStream_t)stream;
    if ((select || both) && side) {
        side_stream = (hipStream_t)side->stream;
        if (hipEventRecord((hipEvent_t)side->fork, (hipStream_t)stream) != hipSuccess) return;
        if (hipStreamWaitEvent(side_stream, (hipEvent_t)side->fork, 0) != hipSuccess) return;
    }
    if (select) hipLaunchKernelGGL(tile_select_kernel, dim3(capped(nq)), dim3(64), 0, side_stream, sc, w);
    if (both) hipLaunchKernelGGL(tile_replay_kernel, dim3(capped((nq + 63) / 64)), dim3(64), 0, side_stream, sc, w, wave_max);
    hipLaunchKernelGGL(tile_replay_wave_kernel, dim3(capped(nq)), dim3(64), 0, (hipStream_t)stream, sc, w, wave_max);
    if ((select || both) && side) {
        if (hipEventRecord((hipEvent_t)side->join, side_stream) != hipSuccess) return;
        if (hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)side->join, 0) != hipSuccess) return;
    }
    hipLaunchKernelGGL(tile_assemble_kernel<false>, dim3(capped(b.n)), dim3(64), assemble_lds, (hipStream_t)stream, sc, b, w);
}
void launch_rescore(const DevDbView& db, const DevScorer& sc, const DevBatchView& b, const DevWork& w,
                    const double* lnfact_table, uint32_t lnfact_n, uint32_t max_ions, SageFeature* out,
                    uint32_t* out_count, uint8_t* keep, void* stream) {
    if (b.n == 0) return;
    if (sc.big_path) {  // report_psms > 32, or peptides of more than 1023 residues
        const uint32_t cap = w.hugebuf ? HUGE_GRID : TILE_GRID_CAP;
        const auto big = w.hugebuf ? (sc.long_runs ? rescore_big_kernel<true, true> : rescore_big_kernel<false, true>)
                                   : (sc.long_runs ? rescore_big_kernel<true, false> : rescore_big_kernel<false, false>);
        hipLaunchKernelGGL(big, dim3(b.n < cap ? b.n : cap), dim3(64),
                           rescore_lds_bytes(sc, b, max_ions, keep != nullptr, w.hugebuf != nullptr), (hipStream_t)stream, db, sc, b, w,
                           lnfact_table, lnfact_n, out, out_count, keep);
        return;
    }
    // the first pass of a two-pass search (sc.fast_log) runs the instance with the logarithm's fast phase only (crlog.h)
    // (the production instance comes in two forms: with the short divisions where the host allows them — DevScorer::tol_mode)
    const auto kern = w.dbg ? rescore_kernel<true, true>
                      : (sc.fast_log && !keep) ? ((sc.tol_mode & TOL_FAST) ? rescore_kernel<false, false, true> : rescore_kernel<false, false>)
                                               : rescore_kernel<false, true>;
    const RescoreKernargs args{db, sc, b, w, lnfact_table, lnfact_n, out, out_count, keep};
    hipLaunchKernelGGL(kern, dim3(b.n), dim3(64), rescore_lds_bytes(sc, b, max_ions, keep != nullptr), (hipStream_t)stream, args);
}
void launch_epilogue(const uint32_t* counts, uint32_t n, uint32_t* h_counts, const uint32_t* order, const EpilogueParts& parts, void* stream) {
    const uint32_t blocks = (n + 1023) / 1024;
    hipLaunchKernelGGL(epilogue_kernel, dim3(blocks ? (blocks < 512u ? blocks : 512u) : 1u), dim3(256), 0, (hipStream_t)stream, counts, n, h_counts,
                       order, parts);
}
// The largest candidate-slot count (scoring.rs:351: right - left + 1 of IndexedDatabase::query, database.rs:402-425) any
// precursor-window query of the batch will see — what decides whether the large-window kernels have to be launched at all
// (slots beyond DevScorer::wcap).  One THREAD per spectrum, two plain binary searches over the peptide masses per query: the
// same partition points in the same total order as query_window's.
__global__ __launch_bounds__(256) void window_max_kernel(DevScorer sc, DevBatchView b, const float* __restrict__ pep_mono, uint32_t np,
                                                         uint32_t* __restrict__ out_max) {
    const uint32_t spec = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t widest = 0;
    if (spec < b.n) {
        const uint32_t zraw = b.precursor_charge[spec];
        const bool ranged = sc.wide_window || zraw == 0 || sc.override_precursor_charge;  // scoring.rs:423, 437, 442
        const uint32_t z0 = ranged ? sc.min_precursor_charge : zraw, z1 = ranged ? sc.max_precursor_charge : zraw;
        const float mzp = b.precursor_mz[spec] - PROTON;
        Tol iso_tol{2, -2.4f, 2.4f};
        if (b.isolation_lo && b.isolation_hi) {
            const float a = b.isolation_lo[spec], c = b.isolation_hi[spec];
            if (a == a && c == c) { iso_tol.lo = a; iso_tol.hi = c; }
        }
        const bool fold = sc.min_isotope_err != sc.max_isotope_err;
        const int isoA = fold ? sc.min_isotope_err : 0, isoB = fold ? sc.max_isotope_err : 0;
        for (uint32_t z = z0; z <= z1 && z != 0; z++) {
            const Tol ptol = sc.wide_window ? tol_scaled(iso_tol, (float)z) : sc.precursor_tol;
            for (int iso = isoA; iso <= isoB; iso++) {
                float plo, phi;
                tol_bounds(ptol, mzp * (float)z - (float)iso * NEUTRON, plo, phi);
                const int32_t klo = order_key(plo), khi = order_key(phi);
                uint32_t a = 0, e = np;  // first index with key >= klo
                while (a < e) {
                    const uint32_t m = a + ((e - a) >> 1);
                    if (order_key(pep_mono[m]) < klo) a = m + 1; else e = m;
                }
                const uint32_t left = a ? a - 1 : 0;
                uint32_t c = left;
                e = np;  // first index with key > khi
                while (c < e) {
                    const uint32_t m = c + ((e - c) >> 1);
                    if (order_key(pep_mono[m]) <= khi) c = m + 1; else e = m;
                }
                const uint32_t potential = c - left + 1;  // (right - left + 1; an inverted window gives <= 2)
                widest = potential > widest ? potential : widest;
            }
        }
    }
    widest = (uint32_t)wave_max_i64((long long)widest);
    if ((threadIdx.x & 63u) == 0 && widest) atomicMax(out_max, widest);
}
void launch_window_max(const DevScorer& sc, const DevBatchView& b, const float* pep_mono, uint32_t np, uint32_t* out_max, void* stream) {
    if (b.n == 0) return;
    hipLaunchKernelGGL(window_max_kernel, dim3((b.n + 255) / 256), dim3(256), 0, (hipStream_t)stream, sc, b, pep_mono, np, out_max);
}
void launch_quick_mark(const DevScorer& sc, const DevBatchView& b, const DevWork& w, uint8_t* keep, void* stream) {
    if (b.n == 0) return;
    hipLaunchKernelGGL(quick_mark_kernel, dim3((b.n + 3) / 4), dim3(256), 0, (hipStream_t)stream, sc, b.n, w, keep);
}
void launch_annotate(const DevDbView& db, const DevScorer& sc, const DevBatchView& b, const SageFeature* feats,
                     const uint32_t* counts, const uint64_t* psm_off, const DevFragments& out, void* stream) {
    if (b.n == 0) return;
    const size_t lds = ((size_t)b.pcap * 10 + 15) & ~(size_t)15;
    hipLaunchKernelGGL(annotate_kernel, dim3(b.n), dim3(64), lds, (hipStream_t)stream, db, sc, b, feats, counts, psm_off, out);
}

}  // namespace sagehip
