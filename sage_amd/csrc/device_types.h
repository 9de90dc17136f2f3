// Plain-data views passed by value to the HIP kernels (kernels.hip) by the C-ABI layer (capi.hip).
#pragma once
#include <stdint.h>

#include <string>

#include "../../include/sage_hip.h"
#include "core.h"

namespace sagehip {

// Device-resident database.
struct DevDbView {
    const float* pep_mono;        // [np]   peptide masses, ascending — the precursor-window search key
    uint32_t np;
    // ... and its position table (index_build.hip: build_peptide_mass_lut): pep_lut[b] = first peptide with mass >= b / pep_lut_inv_w
    // in the total order, b = 0..pep_lut_bins (pep_lut[pep_lut_bins] == np); pep_lut_bins == 0: none
    const uint32_t* pep_lut;
    uint32_t pep_lut_bins;
    float pep_lut_inv_w;
    // peptide-major copy of IndexedDatabase.fragments: the same (peptide_index, fragment_mz) entries grouped by
    // peptide, so that a precursor window is ONE contiguous, coalesced range (narrow kernel, small windows)
    const SageTheoretical* pm_frag;  // [nf]
    const uint64_t* pm_off;          // [np + 1]
    // complete ion table for rescoring: ions[ion_off[p] + k*(L-1) + idx] = IonSeries(p, kinds[k])[idx]
    const float* ions;
    const uint64_t* ion_off;         // [np + 1]
    const uint32_t* pep_info;        // [np] len | decoy<<16 | missed_cleavages<<24
    // tile-major copy of the fragments for large precursor windows: tile = peptide_index >> tile_shift,
    // ascending m/z inside a tile, plus a per-tile position table:
    //   tm_lut[tm_lut_index(t, c)] = position in tm_frag of tile t's first fragment with m/z >= c / lut_scale
    // (the last cell of a tile is its end position), row-major: lut[t][c].  (Round 4 measured the transposed layout, lut[c][t] —
    // the words one fragment-tolerance window needs from the ~100 consecutive tiles of a precursor window share cache lines:
    // same kernel time, 16 % MORE HBM traffic on C4 (2.48 against 2.13 MB per spectrum: a tile's windows no longer share the
    // tile's row) — TM_LUT_TRANSPOSED keeps both layouts buildable.)  A (peak window, tile) lookup is two table reads and a
    // short contiguous run of entries; the tile's candidate counters fit in LDS.
    const SageTheoretical* tm_frag;  // [nf + 2]
    const uint32_t* tm_lut;          // [n_tiles * lut_stride]
    uint32_t tile_shift;
    uint32_t n_tiles;
    uint32_t lut_stride;
    float lut_scale;
    // a second tile-major copy with SMALL tiles (2^tile2_shift peptides, coarser cells) for the narrow kernel's per-peak
    // lookups: a +-10 ppm window holds a few hundred candidates, and a run of a small tile is ~3 entries instead of ~30
    const SageTheoretical* tm2_frag;  // [nf + 2]
    // its position table in succinct form (core.h: LutWord / lut_rank): lut2_words words per tile, each {occupancy bits of 32 cells,
    // rank = index into tm2_pos of the first non-empty cell at or after the word's first cell}; tm2_pos: the run starts of a tile's
    // non-empty cells in cell order, then the tile's end
    const sagecore::LutWord* tm2_l1;  // [n_tiles2 * lut2_words]
    const uint32_t* tm2_pos;          // [non-empty cells of all tiles + n_tiles2]
    uint32_t lut2_words;              // ceil(lut2_stride / 32)
    uint32_t tile2_shift;
    uint32_t n_tiles2;
    uint32_t lut2_stride;
    float lut2_scale;
    uint64_t nf;
    uint8_t ion_kinds[8];
    uint32_t n_kinds;
};

// Layout of the large tiles' position table (tm_lut).  0: row-major, lut[t][c].  1: transposed, lut[c][t] (round 4's experiment:
// the words one window needs from consecutive tiles share lines — and are evicted from L2 before the next tile asks).  2: QUADS,
// lut[t / 4][c][t % 4] (round 5): the words of a cell for four consecutive tiles are one aligned 16-byte load, so the count
// kernel reads a window's table words once per FOUR tiles (kernels.hip: issue_lut) — a quarter of the table's line requests.
// The table has tm_lut_rows(n_tiles) rows; rows beyond n_tiles describe empty tiles.
#ifndef SAGE_TM_LUT_TRANSPOSED
#define SAGE_TM_LUT_TRANSPOSED 0
#endif
#ifndef SAGE_TM_LUT_QUAD
#define SAGE_TM_LUT_QUAD 0
#endif
constexpr int TM_LUT_LAYOUT = SAGE_TM_LUT_QUAD != 0 ? 2 : SAGE_TM_LUT_TRANSPOSED != 0 ? 1 : 0;
constexpr bool TM_LUT_TRANSPOSED = TM_LUT_LAYOUT == 1;
__host__ __device__ inline uint32_t tm_lut_rows(uint32_t n_tiles) { return TM_LUT_LAYOUT == 2 ? (n_tiles + 3u) & ~3u : n_tiles; }
__host__ __device__ inline size_t tm_lut_index(uint32_t t, uint32_t c, uint32_t n_tiles, uint32_t lut_stride) {
    return TM_LUT_LAYOUT == 2 ? (((size_t)(t >> 2) * lut_stride + c) << 2) + (t & 3u)
           : TM_LUT_LAYOUT == 1 ? (size_t)c * n_tiles + t : (size_t)t * lut_stride + c;
}

struct DevScorer {
    sagecore::Tol precursor_tol, fragment_tol;
    sagecore::PbmReach pbm_reach;  // pbm_reach_of(fragment_tol): the peak-presence bitmap's reach coefficients (core.h)
    uint32_t min_matched_peaks;
    int min_isotope_err, max_isotope_err;
    uint32_t min_precursor_charge, max_precursor_charge;
    uint32_t override_precursor_charge;
    int max_fragment_charge;  // -1 == None
    uint32_t chimera;
    uint32_t report_psms;
    uint32_t wide_window;
    int score_type;
    uint32_t kmax;       // max(50, 2*report_psms): upper bound of every trim_k()
    uint32_t big_path;   // 1: the instances for lists wider than a wavefront (report_psms > 32: heaps in LDS, every trim exact,
                         //    rescore_big_kernel) — also taken by a database with peptides of more than 1023 residues, whose
                         //    Run states need the two-register form (long_runs; core.h: run_matched_packed)
    uint32_t long_runs;  // 1: ion indices beyond 1023 occur
    uint32_t tol_mode;   // core.h: TolMode bits — how the rescoring kernels may evaluate Tolerance::bounds of a fragment and the m/z of
                         //    a charge-3 fragment (capi.hip: scorer_tol_mode)
    uint32_t list_cap;   // capacity (entries) of each of the two CLists
    uint32_t wcap;       // candidate-slot capacity of the LDS counter array of the narrow kernel: spectra with a
                         // larger precursor window go to the tiled large-window kernel
    uint32_t dbg_flags;  // timing experiments only (SAGE_HIP_DEBUG_FLAGS)
    uint32_t xcd_chunk;  // consecutive schedule positions one XCD takes at a time (kernels.hip: xcd_position); 0: round-robin
    uint32_t fast_log;   // 1: this pass is followed by the exact retry pass, so its rescoring kernel may carry the fast phase of the
                         //    correctly rounded logarithm only and queue the (rare) spectrum it cannot round for (crlog.h)
    uint32_t exact;      // 1: every trim_hits replays bounded_min_heapify, so the preliminary list has the reference's heap
                         //    layout.  0: each trim keeps the same SET of candidates (the k largest) without replaying the
                         //    heap; the layout is only observable through equal hyperscores at a reported rank, which the
                         //    rescoring kernel detects and sends back through the exact path
};

struct DevBatchView {
    uint32_t n;                 // spectra of this batch == upper bound of the launch grids
    const uint32_t* n_dev;      // when not null: the number of entries of `order` to score lives on the device (the exact
                                //     retry pass is launched without a host round trip: its count is a counter of pass 1)
    uint32_t spec_base;         // added to Feature.spec_index: position of this batch's first spectrum in the caller's batch
    const uint64_t* peak_off;
    const float* masses;
    const float* intensities;
    const float* precursor_mz;
    const uint8_t* precursor_charge;
    const float* isolation_lo;  // may be null
    const float* isolation_hi;
    const float* tic;
    const float* rt;            // may be null
    const float* ims;           // may be null
    const uint32_t* file_id;    // may be null
    const uint32_t* order;      // [n] block b scores spectrum order[b]: ascending precursor mass => neighbouring
                                //     wavefronts stream overlapping index ranges (L2 reuse); results stay in input order
    uint32_t probe;             // narrow kernel variant: 1 = per-peak table lookups (large windows), 0 = peptide-major stream
    uint32_t pcap;              // max peaks per spectrum in this batch
    uint32_t fzcap;             // max (max_fragment_charge - 1) over the charges this batch can use
    const uint4* sched;         // may be null.  [2 n] what a block needs of the spectrum it scores, IN SCHEDULE ORDER: record b =
                                //     {order[b], peaks, peak_off lo, hi}, {charge, precursor m/z, isolation lo, hi (NaN: none)} — one
                                //     trip to a line its neighbours share instead of order[b] and then five random reads
};

struct DevWork {  // per-spectrum outputs of the preliminary pass
    uint64_t* cand;        // [n * kmax] packed PreScore in the reference's heap-layout order
    uint32_t* cand_len;    // [n]
    uint32_t* totals;      // [n * 2] InitialHits.matched_peaks, .scored_candidates
    uint32_t* status;      // [n] 0 ok, 1 deferred to the large-window kernel, 2 list overflow
    uint32_t* n_deferred;  // [8]: [0] spectra queued for the large-window kernels, [1] list overflows,
                           //      [2] queue head (next entry a large-window workgroup takes),
                           //      [3] candidate-arena bump pointer, [4] candidate-arena overflows
    uint32_t* queue;       // [n] spectra queued for the large-window kernels
    uint32_t* retry;       // [n] spectra whose reported ranks tie in hyperscore: re-run with exact heap layouts
    uint32_t tile_blocks;  // persistent workgroups of the large-window counting kernel
    uint32_t cnt8;         // count in u8 (first pass of a two-pass search only: overflowing spectra go to the u16 retry pass)
    uint32_t reuse;        // retry pass: the large-window counts of the first pass are reused through item_of (kernels.hip: query_slot)
    uint32_t queue_later;  // prelim_kernel only marks a spectrum it hands to the large-window kernels (status); queue_kernel appends the
                           //     marked ones to `queue` behind it, a wavefront's worth per atomic (kernels.hip: queue_kernel)
    uint32_t* item_of;     // [n] spectrum -> its item in the first pass's queue; bit 31: count again (a u8 counter may have wrapped)
    uint32_t* ready;       // [n] search_kernel: == epoch once the spectrum's preliminary list is in HBM (written with agent-scope release)
    uint32_t epoch;        //     of this launch (never 0; the array is zeroed when it is allocated)
    uint64_t replay_split; // which heap-replay kernel takes a query: low 32 bits = queries up to which every query gets a wavefront
                           //     (SAGE_HIP_REPLAY_WAVE_MAX; default 0xFFFFFFFF: every query, the lane-per-query kernel is not launched), high 32 bits = stream words above which a query does
                           //     anyway (SAGE_HIP_REPLAY_LANE_MAX, 0: the default of kernels.hip)
    uint32_t kstride;      // entries per query of `seeds` / `qres`: kmax rounded up to a multiple of 64 (64 unless report_psms > 32)
    uint32_t search_lag;   //     workgroup ids by which a spectrum's rescoring trails its preliminary workgroup (0: the default)
    // Cheap ties (kernels.hip: rescore_spectrum).  The preliminary kernel leaves the window counts of every narrow single-query
    // spectrum in HBM; where the hyperscores of a spectrum's best candidates tie (and report_psms == 1, no chimera), the rescoring
    // wavefront replays bounded_min_heapify from them on the spot, finds which of the tied candidates comes first in the
    // reference's preliminary list and reports it — instead of queueing the spectrum for the exact retry pass.
    uint32_t* cnt_store;   // [n * cnt_stride] one row per SCHEDULE POSITION of the launch (null: off): {left, potential, -, -} of the
                           //     spectrum's query — potential == 0: nothing kept (several queries, large window) — then its u16
                           //     counts in slot order, two per word
    uint32_t cnt_stride;   //     words per row (4 + wcap / 2, a multiple of 4)
    uint32_t* arena_ptr;   // the arena's bump pointer (the first pass's counter in both passes when the retry pass reuses its candidates)
    // large-window pipeline (count -> replay -> assemble); query id = queue position * qmax + (z - z0) * n_iso + iso index
    struct QueryRec* qrec; // [n * qmax]
    uint16_t* seeds;       // [n * qmax * kstride] matched counts of the first min(k, potential) candidate slots
    uint64_t* qres;        // [n * qmax * kstride] heap of each k-selected query, in the reference's layout order
    unsigned char* hugebuf;  // non-null: the lists / heaps / per-candidate arrays of the wide-list kernels (report_psms > 32) in global
    uint32_t huge_stride;    //     memory, [huge_grid()] slices of huge_stride bytes, one per workgroup (they no longer fit a CU's LDS)
    float* winbuf;         // non-null: tile_count_wing_kernel — [tile_blocks][2][fzcap * pcap] the spectrum's windows in global memory
    uint32_t* arena;       // candidate segments: {next, n, tile_base, 0} then n entries `count << 16 | slot in tile`
    uint32_t arena_cap;    // entries
    uint32_t qmax;
    uint32_t tile_shift;   // log2 of the peptides per tile (the readers of the candidate directories need it)
    unsigned long long* dbg;  // optional [2][8] per-phase cycle accumulators (null in production)
};

enum { ST_OK = 0, ST_DEFERRED = 1, ST_OVERFLOW = 2, ST_RETRY = 3,
       ST_DONE = 4,
       ST_OK_ORDERED = 5 };  // ST_OK, and no trim_hits of the spectrum dropped anything: the order-free list IS the reference's list  // reported by the fused narrow kernel: nothing left to do for the per-phase kernels of the same pass
enum { CTR_QUEUED = 0, CTR_LIST_OVERFLOW = 1, CTR_QUEUE_HEAD = 2, CTR_ARENA_PTR = 3, CTR_ARENA_OVERFLOW = 4, CTR_RETRY = 5,
       CTR_TIED = 6,  // narrow spectra whose tie at a reported rank the fused kernel settled in place
       CTR_FAST_TIE = 7,  // (unused since round 6: the stripes below)
       // spectra whose tie at the top rescore_kernel settled from the stored window counts (statistics: SageTiming::n_tied), in
       // CTR_TIE_STRIPES counters 64 bytes apart, a workgroup's by its id: one counter took one atomic per tied spectrum on ONE
       // address — 312 816 per step of the tie-rich C3T, and same-address atomics retire at ~87 M/s: 5.5 % of its rescoring kernel
       CTR_TIE_STRIPE0 = 16, CTR_TIE_STRIPES = 16, CTR_TIE_STRIDE = 16,
       CTR_COUNT = CTR_TIE_STRIPE0 + CTR_TIE_STRIPES * CTR_TIE_STRIDE };

// one precursor-window query (scoring.rs:335-382) of a spectrum handled by the large-window pipeline
struct QueryRec {
    uint32_t left;        // pre_idx_lo: candidate slot s <-> peptide left + s
    uint32_t potential;   // number of candidate slots (scoring.rs:351); 0 == query not evaluated
    uint32_t matched;     // InitialHits.matched_peaks of this query
    uint32_t scored;      // InitialHits.scored_candidates of this query
    uint32_t head;        // the query's candidate directory in the arena (0xFFFFFFFF: none): n_dir entries {position, count}
    uint32_t z_iso;       // precursor charge | (isotope error + 128) << 8
    uint32_t pad[2];      // [0] bit 0: some count >= 63 (clipped histogram), bits 8..: k-th largest count T; [1] slots == T to skip
    uint32_t n_dir;       // directory entries: tiles of the window x wavefronts of the count kernel, in slot order
    uint32_t t0;          // first tile of the window
    uint32_t n_cand;      // candidate words behind the directory (the length of the stream a heap replay walks)
};

// device-side SageFragments (sage_hip.h)
struct DevFragments {
    uint64_t capacity;
    uint8_t* kinds;
    int32_t* charges;
    int32_t* fragment_ordinals;
    float* intensities;
    float* mz_calculated;
    float* mz_experimental;
};

struct TileParams {
    DevDbView db;
    DevScorer sc;
    DevBatchView b;
    DevWork w;
};

// launch wrappers (kernels.hip)
size_t prelim_lds_bytes(const DevScorer& sc, const DevBatchView& b, bool huge = false);
size_t huge_stride_bytes(const DevScorer& sc);  // bytes of DevWork::hugebuf per workgroup of the wide-list kernels
uint32_t huge_grid();
size_t tile_lds_bytes(const DevDbView& db, const DevScorer& sc, const DevBatchView& b, bool cnt8 = false, bool wing = false);
int tile_kernel_prepare(size_t max_lds_bytes);  // raises the kernel's dynamic-LDS limit; returns a hipError_t
int spectrum_kernel_prepare(size_t max_lds_bytes);  // ... of the per-spectrum kernels (spectra of thousands of peaks)
int bigk_kernel_prepare(size_t max_lds_bytes);  // ... of the instances for lists wider than a wavefront (report_psms > 32)
size_t assemble_lds_bytes(const DevScorer& sc);
uint32_t fast_tie_lds_words();  // words of LDS rescore_kernel can stage a spectrum's window counts in ((wcap + 1) / 2 must fit)
size_t rescore_lds_bytes(const DevScorer& sc, const DevBatchView& b, uint32_t max_ions, bool quick, bool huge = false);
size_t narrow_lds_bytes(const DevScorer& sc, const DevBatchView& b);
// the narrow search as ONE launch of two kinds of workgroups (preliminary / rescoring, kernels.hip: search_kernel)
size_t search_lds_bytes(const DevScorer& sc, const DevBatchView& b);
void launch_search(const DevDbView& db, const DevScorer& sc, const DevBatchView& b, const DevWork& w, const double* lnfact_table,
                   uint32_t lnfact_n, SageFeature* out, uint32_t* out_count, void* stream);
// the fused narrow-search kernel: preliminary matching + k-select + rescoring of every spectrum whose windows fit the LDS counters
void launch_narrow(const DevDbView& db, const DevScorer& sc, const DevBatchView& b, const DevWork& w, const double* lnfact_table,
                   uint32_t lnfact_n, SageFeature* out, uint32_t* out_count, void* stream);
void launch_prelim(const DevDbView& db, const DevScorer& sc, const DevBatchView& b, const DevWork& w, void* stream);
void launch_queue(const DevScorer& sc, const DevBatchView& b, const DevWork& w, void* stream);  // (DevWork::queue_later)
// `side`: a second stream + two events (hipStream_t, hipEvent_t x 2) for the launches that may run next to each other, or null
struct SideStream {
    void* stream;
    void* fork;
    void* join;
};
void launch_prelim_tile(const DevDbView& db, const DevScorer& sc, const DevBatchView& b, const DevWork& w, void* stream,
                        const SideStream* side = nullptr);
uint32_t queries_per_spectrum(const DevScorer& sc);
// index_build.hip (both return a hipError_t)
int generate_fragments_on_device(uint64_t np, uint32_t nk, const uint8_t* d_kinds, const uint64_t* d_seq_off, const uint8_t* d_seq,
                                 const float* d_mods, const float* d_nterm, const float* d_mono, uint64_t min_ion_index,
                                 const uint64_t* d_ion_off, const uint64_t* d_pm_off, float* d_ions, SageTheoretical* d_pm_frag,
                                 void* stream);
int ion_abs_range_on_device(const float* d_ions, uint64_t n, uint32_t* lo_bits, uint32_t* hi_bits);
int rebuild_peptide_major_on_device(const SageTheoretical* d_tm_frag, uint64_t nf, SageTheoretical* d_pm_frag, void* stream);
int build_tile_copy_on_device(const SageTheoretical* d_pm_frag, uint64_t nf, uint32_t tile_shift, uint32_t n_tiles,
                              const uint64_t* d_tile_off, float lut_scale, SageTheoretical* d_tm_frag, uint32_t** d_lut_out,
                              uint32_t* lut_stride_out, void* stream, int layout = 0);
// lut[n_tiles][lut_stride] (row-major, as build_tile_copy_on_device makes it) -> its succinct form; the arrays are allocated here
int build_succinct_lut_on_device(const uint32_t* d_lut, uint32_t n_tiles, uint32_t lut_stride, sagecore::LutWord** d_l1_out, uint32_t** d_pos_out,
                                 uint32_t* words_out, uint64_t* n_pos_out, void* stream);
int build_peptide_mass_lut(const float* d_pep_mono, uint32_t np, float top_mass, uint32_t** d_lut_out, uint32_t* bins_out, float* inv_w_out,
                           void* stream);
// rescore.hip
int rescore_on_device(int device, const SageRescoreInput& in, SageRescoreOutput& out, std::string& err);
int predict_rt_on_device(int device, const SageRtInput& in, SageRtOutput& out, std::string& err);
// process.hip
// launch schedule of a batch (index_build.hip): order[k] = spectrum scored by block k, ascending neutral precursor mass
size_t schedule_temp_bytes(uint32_t n);
int schedule_on_device(uint32_t n, const float* d_precursor_mz, const uint8_t* d_charge, uint32_t min_charge, uint32_t* d_keys_a,
                       uint32_t* d_keys_b, uint32_t* d_idx, uint32_t* d_order, void* d_temp, size_t temp_bytes, void* stream);
void schedule_records_on_device(uint32_t n, const uint32_t* d_order, const uint64_t* d_peak_off, const float* d_precursor_mz,
                                const uint8_t* d_charge, const float* d_iso_lo, const float* d_iso_hi, uint4* d_sched, void* stream);
size_t process_lds_bytes(uint32_t rcap, uint32_t rpow2);
int process_kernel_prepare(size_t max_lds_bytes);
void launch_process(uint32_t n, const uint64_t* raw_off, const float* raw_mz, const float* raw_int, const uint8_t* charge,
                    uint32_t take_top_n, bool deisotope, float min_deisotope_mz, uint32_t rcap, uint32_t rpow2, uint32_t stride,
                    float* out_mass, float* out_int, float* out_tic, uint32_t* out_count, void* stream);
void launch_process_big(uint32_t n_big, const uint32_t* big_list, unsigned char* workspace, const uint64_t* raw_off, const float* raw_mz,
                        const float* raw_int, const uint8_t* charge, uint32_t take_top_n, bool deisotope, float min_deisotope_mz,
                        uint32_t big_cap, uint32_t big_pow2, uint32_t stride, float* out_mass, float* out_int, float* out_tic,
                        uint32_t* out_count, void* stream);
void launch_compact(uint32_t n, const uint64_t* peak_off, uint32_t stride, const float* sm, const float* si, float* masses,
                    float* intens, void* stream);
void launch_rescore(const DevDbView& db, const DevScorer& sc, const DevBatchView& b, const DevWork& w,
                    const double* lnfact_table, uint32_t lnfact_n, uint32_t max_ions, SageFeature* out,
                    uint32_t* out_count, uint8_t* keep, void* stream);
// counts[n] -> h_counts (device view of page-locked memory) and the counter blocks (2 * CTR_COUNT words) of up to four parts
struct EpilogueParts {
    uint32_t* src[4];  // (reset by the kernel after the copy)
    uint32_t* dst[4];
    uint32_t n;
};
void launch_epilogue(const uint32_t* counts, uint32_t n, uint32_t* h_counts, const uint32_t* order, const EpilogueParts& parts, void* stream);
void launch_window_max(const DevScorer& sc, const DevBatchView& b, const float* pep_mono, uint32_t np, uint32_t* out_max, void* stream);
void launch_quick_mark(const DevScorer& sc, const DevBatchView& b, const DevWork& w, uint8_t* keep, void* stream);
void launch_annotate(const DevDbView& db, const DevScorer& sc, const DevBatchView& b, const SageFeature* feats,
                     const uint32_t* counts, const uint64_t* psm_off, const DevFragments& out, void* stream);

}  // namespace sagehip
