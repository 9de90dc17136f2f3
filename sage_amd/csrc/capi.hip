// capi.hip — the C ABI declared in include/sage_hip.h.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>
#include <cstring>
#include <memory>
#include <future>
#include <mutex>
#include <string>
#include <vector>

#include "crlog.h"
#include "device_types.h"
#include "host_db.hpp"

using namespace sagehip;

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(_e == hipErrorOutOfMemory ? SAGE_HIP_ERR_OOM : SAGE_HIP_ERR_HIP,           \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                        \
    } while (0)

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    hipError_t alloc(size_t count) {
        release();
        n = count;
        return hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T));
    }
    hipError_t upload(const T* src, size_t count) {
        hipError_t e = alloc(count);
        if (e != hipSuccess) return e;
        return count ? hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice) : hipSuccess;
    }
    // grow-only: keep the allocation when it is large enough (no driver call on the steady-state path)
    hipError_t reserve(size_t count) {
        if (p && count <= n) return hipSuccess;
        return alloc(std::max(count, n + n / 2));
    }
    size_t bytes() const { return n * sizeof(T); }
};

// page-locked host block, grow-only (staging of the streaming pipeline: DMA at full PCIe rate)
struct Pinned {
    unsigned char* p = nullptr;
    size_t cap = 0;
    Pinned() = default;
    Pinned(const Pinned&) = delete;
    Pinned& operator=(const Pinned&) = delete;
    ~Pinned() {
        if (p) (void)hipHostFree(p);
    }
    hipError_t reserve(size_t bytes) {
        if (p && bytes <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 4 + 256;
        const hipError_t e = hipHostMalloc((void**)&p, want, hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
};

template <class T>
T* carve(unsigned char*& cur, size_t count) {  // next 64-byte aligned array of a staging block
    T* p = (T*)cur;
    cur += ((count * sizeof(T) + 63) / 64) * 64;
    return p;
}

struct Event {
    hipEvent_t e = nullptr;
    Event() = default;
    Event(const Event&) = delete;
    Event& operator=(const Event&) = delete;
    ~Event() {
        if (e) (void)hipEventDestroy(e);
    }
    hipError_t create(bool timing) { return e ? hipSuccess : hipEventCreateWithFlags(&e, timing ? hipEventDefault : hipEventDisableTiming); }
};

}  // namespace

struct SageHostDb {
    HostDb db;
};

struct SageDeviceDb {
    int device = 0;
    DevBuf<float> pep_mono;
    DevBuf<uint32_t> pep_lut;  // position table of pep_mono (DevDbView::pep_lut)
    DevBuf<float> ions;
    DevBuf<uint64_t> ion_off;
    DevBuf<uint32_t> pep_info;
    // peptide-major copy: the source of the tile copies while the index is built, afterwards read by the STREAM variant of the
    // preliminary kernels only (windows of a handful of candidates) — released when the build is done (1.16 GB of C3's HBM) and
    // made again, from a tile copy, by the first batch that takes the stream variant (ensure_pm_frag; SAGE_HIP_KEEP_PM_FRAG=1 keeps it)
    DevBuf<SageTheoretical> pm_frag;
    std::mutex pm_mu;
    DevBuf<uint64_t> pm_off;
    std::vector<float> h_pep_mono;    // host copy: window-size estimate at batch upload
    DevBuf<SageTheoretical> tm_frag;  // tile-major copy + position table for the large-window kernel
    DevBuf<uint32_t> tm_lut;
    DevBuf<SageTheoretical> tm2_frag; // small-tile copy + table for the narrow kernel's per-peak lookups
    DevBuf<sagecore::LutWord> tm2_l1;  // its position table in succinct form (core.h: LutWord): occupancy + rank per 32 cells ...
    DevBuf<uint32_t> tm2_pos;          // ... and the run starts of the non-empty cells
    uint32_t max_ions = 0;
    uint32_t max_len = 0;   // residues of the longest peptide (> 1023: DevScorer::long_runs)
    uint32_t ion_lo_bits = 0xFFFFFFFFu, ion_hi_bits = 0;  // smallest / largest |ion| of the rescoring table, as f32 bits (scorer_tol_mode)
    DevDbView view{};
    uint64_t bytes = 0;
};

// Device-side working set of ONE scoring pass pair (device_types.h: DevWork).  Kernels of successive batches run in order on
// one compute stream, so a scorer needs a single set whatever the number of batches in flight.
struct WorkSet {
    DevBuf<uint64_t> cand;
    DevBuf<uint32_t> cand_len, totals, status, queue, retry, item_of, ready;
    uint32_t epoch = 0;  // of the last search_kernel launch (DevWork::epoch)
    DevBuf<QueryRec> qrec;
    DevBuf<uint16_t> seeds;
    DevBuf<uint64_t> qres;
    DevBuf<uint32_t> arena;
    // cheap ties (device_types.h: DevWork::cnt_store): the window counts of every narrow single-query spectrum
    DevBuf<uint32_t> cnt_store;
    DevBuf<float> winbuf;   // tile_count_wing_kernel's windows (DevWork::winbuf), when a batch needs them
    DevBuf<unsigned char> hugebuf;  // the wide-list kernels' lists in global memory (DevWork::hugebuf), when a configuration needs them
    uint32_t cap_tie = 0;
    uint32_t cap_n = 0;     // spectra the narrow-path buffers hold
    uint32_t cap_wide = 0;  // spectra the large-window buffers hold (0 on the second compute lane, always)
};

// What a batch leaves behind: PSM records, counters, timing events.  Double-buffered by the streaming pipeline (batch c + 1 is
// scored while the records of batch c cross PCIe).
struct OutSet {
    DevBuf<SageFeature> features;
    DevBuf<uint32_t> out_count;
    DevBuf<uint32_t> counters;       // [2 * CTR_COUNT]: first pass, exact retry pass
    uint32_t* h_counters = nullptr;  // pinned [2 * CTR_COUNT]
    uint32_t* h_counters_view = nullptr;  // ... as the kernels address it
    bool counters_clean = false;  // the last command on the counters was the epilogue kernel's reset (and the host waited for it)
    Pinned h_out;                    // landing block for the records when the caller's arrays are pageable
    Event ev[5];                     // start, prelim 1, rescore 1, prelim 2, rescore 2
    Event comp_done, down_done;
    bool in_flight = false, two_pass = false, with_rescore = false;
    bool fused = false;          // the narrow spectra were scored by the fused kernel (ev[0] -> ev[1])
    bool wide_launched = true;   // the large-window kernels ran behind it (else: the batch was expected to hold narrow windows only)
    bool timed = true;           // ev[] were recorded for this launch (SageScorer::timing_every)
    bool retry_deferred = false; // two_pass, but the retry pass has not been launched (score_resident_locked: phase 1)
    bool huge = false;           // this launch's wide-list kernels kept their lists in WorkSet::hugebuf
    uint32_t n = 0;
    ~OutSet() {
        if (h_counters) (void)hipHostFree(h_counters);
    }
};

struct SageDeviceBatch {
    int device = 0;
    uint32_t n = 0;
    DevBuf<uint64_t> peak_off;
    DevBuf<float> masses, intensities, precursor_mz, iso_lo, iso_hi, tic, rt, ims;
    DevBuf<uint8_t> charge;
    DevBuf<uint32_t> file_id, order, sort_a, sort_b, sort_idx;
    DevBuf<uint8_t> sort_tmp;
    DevBuf<uint4> sched;     // DevBatchView::sched (SAGE_HIP_NO_SCHED=1: not built — the kernels go through `order`)
    DevBuf<uint8_t> meta;    // streaming pipeline: the per-spectrum arrays in one block, the image of the staging block (one copy)
    uint32_t widest = 0xFFFFFFFFu;  // candidate slots of the batch's widest precursor window (exact_window_check; else unknown)
    bool maybe_wide = true;  // some precursor window may exceed the narrow kernel's LDS counters (estimated at upload; a wrong
                             // "no" is noticed after the step — the queue counter — and the step is repeated with the
                             // large-window kernels)
    Pinned stage;      // host staging of the arrays above (everything but the peaks when those are already page-locked)
    Event up_done;     // uploads of this batch finished (its staging block may be refilled)
    Event sort_done;   // the per-spectrum block is up and the launch schedule sorted (on the scorer's sort stream, beside the peak copies)
    DevBatchView view{};
};

struct SageScorer {
    SageDeviceDb* db = nullptr;
    SageScorerParams params{};
    DevScorer dev{};
    std::mutex mu;                   // entry points taking this handle serialise on it (clone the scorer for concurrency)
    hipStream_t stream = nullptr;    // compute (and, for resident batches, the result download)
    hipStream_t up_stream = nullptr, down_stream = nullptr;  // streaming pipeline: H2D of batch c + 1, D2H of batch c - 1
    hipStream_t side_stream = nullptr;  // kernels of one batch that may run next to each other (the two heap-replay kernels); also the
                                        // per-spectrum block + launch-schedule sort of an UPLOADING batch, beside its peak copies
                                        // (stage_and_upload).  Not a stream of its own for that: one more stream per scorer changed
                                        // how HIP maps streams to hardware queues and the two parts of a small resident step
                                        // (way streams) stopped overlapping — 62 500 C3 spectra: 0.79 -> 0.98 ms.
    Event side_fork, side_join;
    // a resident narrow-search step in `ways` parts, each with its own stream (part 0: `stream`): the parts' kernels overlap each
    // other's cold starts, tails and retry chains (what two scorer handles on two host threads get, inside one call)
    uint32_t ways = 0;               // SAGE_HIP_WAYS=1..4; 0: by batch size (score_resident_locked).  Measured on C3: two parts
                                     // -1 % wall at 500 000 spectra, -6 % at 62 500, nothing more with three or four
    hipStream_t way_stream[3] = {nullptr, nullptr, nullptr};
    Event way_fork, way_join[3], way_begin, way_end[4];  // (way_end[p]: behind the last command of part p)
    DevBuf<uint32_t> win_max;   // exact_window_check's result word
    bool retry_likely = true;   // the last resident step had spectra to retry (or there was none yet): the next one launches its
                                // retry pass unconditionally
    // per-kernel HIP events of a step (SageTiming::prelim_ms / rescore_ms / retry_ms): on every `timing_every`-th scoring call
    // (sage_hip_scorer_set_timing_interval; 1: every call, 0: never).  Eight event records and six elapsed-time queries cost a
    // 0.65 ms step ~15 us; a step without them reports the last timed step's kernel times.
    uint32_t timing_every = 1;
    uint64_t timing_calls = 0;
    bool timed = true;          // this call records events
    float kept_prelim_ms = 0.f, kept_rescore_ms = 0.f, kept_retry_ms = 0.f;
    DevBuf<double> lnfact;
    DevBuf<unsigned long long> dbg;  // SAGE_HIP_PHASE_CLOCKS=1: per-phase cycle accumulators
    uint32_t tile_blocks = 0;        // persistent workgroups of the large-window kernel (u16 counters)
    uint32_t tile_blocks8 = 0;       // ... of its u8 instance (smaller LDS footprint: more of them per CU)
    bool cnt8 = true;                // first pass of a search counts in u8 (SAGE_HIP_NO_U8=1 turns it off)
    bool reuse_counts = true;        // the retry pass reuses the first pass's large-window counts (SAGE_HIP_NO_REUSE=1 turns it off)
    SageTiming timing{};
    bool exact_always = false;  // SAGE_HIP_EXACT=1: never use the order-free trims
    bool one_launch = false;    // SAGE_HIP_ONE_LAUNCH=1: the first pass of narrow windows as one launch of two kinds of workgroups
                                // (kernels.hip: search_kernel) instead of prelim_kernel, then rescore_kernel — measured slower, like
                                // the fused kernel: the larger kernel body costs scalar-register spills
    uint32_t kstride = 64;           // entries per query of the large-window pipeline's seed / heap arrays (DevWork::kstride)
    bool two_lanes = true;           // streaming pipeline: two chunks of a narrow batch side by side (SAGE_HIP_ONE_LANE=1: one at a time)
    uint32_t search_lag = 0;         // SAGE_HIP_SEARCH_LAG (DevWork::search_lag)
    bool sched_desc_forced = false;  // SAGE_HIP_SCHED_DESC was given (else: heaviest precursors first for narrow batches only)
    uint64_t replay_split = 0xFFFFFFFFull;  // SAGE_HIP_REPLAY_WAVE_MAX | SAGE_HIP_REPLAY_LANE_MAX << 32 (DevWork::replay_split); default: every query by wavefront
    bool fused = false;         // SAGE_HIP_FUSED=1: the first pass of narrow windows through the fused kernel as well (measured slower
                                // than the two kernels on MI355X — register pressure, DESIGN.md 4.7 — kept for that comparison)
    bool zero_copy = true;      // records go straight to page-locked result arrays (SAGE_HIP_NO_ZEROCOPY=1: device buffer + copy)
    bool fast_ties = true;      // one reported PSM, no chimera: ties at the top settled by rescore_kernel from stored window counts
                                // (SAGE_HIP_NO_FAST_TIES=1: every tie through the exact retry pass, as in round 3)
    uint32_t cnt_stride = 0;    // words per spectrum of WorkSet::cnt_store
    uint32_t qmax = 1;
    WorkSet ws;                 // the working set (lane 0)
    WorkSet ws2;                // a second one: the streaming pipeline scores two chunks of a narrow batch side by side (lane 1)
    OutSet outs[4];             // [k % 4]: the slots of the streaming pipeline; [0..ways): the concurrent parts of a resident step
    SageDeviceBatch slots[4];   // input buffers of the streaming pipeline (chunk k in slot k % 4: uploads run ahead of the kernels)
    uint32_t chunk = 131072;    // spectra per pipeline stage (SAGE_HIP_CHUNK): 39.9 M spectra/s host to host on C3 against 38.2 M at 65 536 and 37.3 M at 262 144
    // scratch of sage_hip_annotate_resident / sage_hip_quick_score_resident, grow-only
    DevBuf<SageFeature> an_feats;
    DevBuf<uint32_t> an_counts;
    DevBuf<uint64_t> an_off;
    DevBuf<uint8_t> an_kinds, keep;
    DevBuf<int32_t> an_charges, an_ord;
    DevBuf<float> an_int, an_calc, an_exp;
};

extern "C" {

// post-search rescoring (rescore.hip)
int sage_hip_rescore(int device, const SageRescoreInput* in, SageRescoreOutput* out) {
    if (!in || !out) return fail(SAGE_HIP_ERR_INVALID, "sage_hip_rescore: null argument");
    if (sage_hip_device_count() <= 0) return fail(SAGE_HIP_ERR_NO_DEVICE, "sage_hip_rescore: no HIP device (there is no CPU fallback)");
    if (in->n >= (1ull << 31)) return fail(SAGE_HIP_ERR_UNSUPPORTED, "sage_hip_rescore: more than 2^31 features");
    if (in->precursor_tol.kind != SAGE_TOL_PPM && in->precursor_tol.kind != SAGE_TOL_DA)
        return fail(SAGE_HIP_ERR_INVALID, "sage_hip_rescore: Pct tolerance should never be used on mz");  // linear_discriminant.rs:142
    out->passing_spectrum = out->passing_peptide = out->passing_protein = 0;
    out->lda_fitted = 0;
    out->device_ms = 0.0f;
    std::memset(out->coef, 0, sizeof(out->coef));
    if (in->n == 0) return SAGE_HIP_OK;
    if (!in->features || !in->peptide_key || !in->protein_key || !out->discriminant_score || !out->posterior_error ||
        !out->spectrum_q || !out->peptide_q || !out->protein_q)
        return fail(SAGE_HIP_ERR_INVALID, "sage_hip_rescore: null array");
    std::string err;
    const int rc = rescore_on_device(device, *in, *out, err);
    return rc == SAGE_HIP_OK ? rc : fail(rc, err);
}

int sage_hip_predict_rt(int device, const SageRtInput* in, SageRtOutput* out) {
    if (!in || !out) return fail(SAGE_HIP_ERR_INVALID, "sage_hip_predict_rt: null argument");
    if (sage_hip_device_count() <= 0) return fail(SAGE_HIP_ERR_NO_DEVICE, "sage_hip_predict_rt: no HIP device (there is no CPU fallback)");
    if (in->n >= (1ull << 31)) return fail(SAGE_HIP_ERR_UNSUPPORTED, "sage_hip_predict_rt: more than 2^31 features");
    out->rt_fitted = out->ims_fitted = 0;
    out->rt_r2 = out->ims_r2 = 0.0;
    out->device_ms = 0.0f;
    if (in->n_files == 0 && in->n) return fail(SAGE_HIP_ERR_INVALID, "sage_hip_predict_rt: n_files == 0");
    if (in->n == 0) {
        for (uint32_t k = 0; out->alignments && k < in->n_files; ++k) out->alignments[k] = SageAlignment{k, 0.0f, 1.0f, 0.0f};
        return SAGE_HIP_OK;
    }
    if (!in->features || !in->seq_off || !in->seq || !in->monoisotopic || !out->spectrum_q || !out->aligned_rt ||
        !out->predicted_rt || !out->delta_rt_model || !out->predicted_ims || !out->delta_ims_model)
        return fail(SAGE_HIP_ERR_INVALID, "sage_hip_predict_rt: null array");
    std::string err;
    const int rc = predict_rt_on_device(device, *in, *out, err);
    return rc == SAGE_HIP_OK ? rc : fail(rc, err);
}

const char* sage_hip_last_error(void) { return g_last_error.c_str(); }
int sage_hip_abi_version(void) { return SAGE_HIP_ABI_VERSION; }

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
int sage_hip_hostdb_build(const char* fasta_text, const SageDbParams* params, SageHostDb** out) {
    if (!fasta_text || !params || !out) return fail(SAGE_HIP_ERR_INVALID, "null argument");
    try {
        auto h = std::make_unique<SageHostDb>();
        h->db = build_database(fasta_text, config_from_params(*params));
        *out = h.release();
        return SAGE_HIP_OK;
    } catch (const std::exception& e) {
        return fail(SAGE_HIP_ERR_INVALID, e.what());
    }
}
int sage_hip_hostdb_build_chunk(const char* fasta_text, const SageDbParams* params, uint64_t first_target, uint64_t n_targets,
                                SageHostDb** out) {
    if (!fasta_text || !params || !out) return fail(SAGE_HIP_ERR_INVALID, "null argument");
    try {
        auto h = std::make_unique<SageHostDb>();
        h->db = build_database(fasta_text, config_from_params(*params), first_target, n_targets);
        *out = h.release();
        return SAGE_HIP_OK;
    } catch (const std::exception& e) {
        return fail(SAGE_HIP_ERR_INVALID, e.what());
    }
}
int sage_hip_fasta_num_targets(const char* fasta_text, const SageDbParams* params, uint64_t* out) {
    if (!fasta_text || !params || !out) return fail(SAGE_HIP_ERR_INVALID, "null argument");
    *out = fasta_num_targets(fasta_text, config_from_params(*params));
    return SAGE_HIP_OK;
}
int sage_hip_prefilter_chunk_size(const char* fasta_text, const SageDbParams* params, uint64_t requested, uint64_t* out) {
    if (!fasta_text || !params || !out) return fail(SAGE_HIP_ERR_INVALID, "null argument");
    *out = prefilter_chunk_size(fasta_text, config_from_params(*params), requested);
    return SAGE_HIP_OK;
}
int sage_hip_hostdb_merge_kept(const SageHostDb* const* chunks, const uint8_t* const* keep, uint32_t n_chunks,
                               const SageDbParams* params, SageHostDb** out) {
    if ((n_chunks && (!chunks || !keep)) || !params || !out) return fail(SAGE_HIP_ERR_INVALID, "null argument");
    try {
        std::vector<const HostDb*> dbs;
        std::vector<const uint8_t*> masks;
        for (uint32_t c = 0; c < n_chunks; ++c) {
            if (!chunks[c] || !keep[c]) return fail(SAGE_HIP_ERR_INVALID, "null chunk");
            dbs.push_back(&chunks[c]->db);
            masks.push_back(keep[c]);
        }
        auto h = std::make_unique<SageHostDb>();
        h->db = merge_kept(dbs, masks, config_from_params(*params));
        *out = h.release();
        return SAGE_HIP_OK;
    } catch (const std::exception& e) {
        return fail(SAGE_HIP_ERR_INVALID, e.what());
    }
}
void sage_hip_hostdb_free(SageHostDb* db) { delete db; }
int sage_hip_hostdb_view(const SageHostDb* db, SageDbView* out) {
    if (!db || !out) return fail(SAGE_HIP_ERR_INVALID, "null argument");
    *out = db->db.view();
    return SAGE_HIP_OK;
}
static uint64_t copy_out(const std::string& s, char* out, uint64_t cap) {
    if (out && cap >= s.size() + 1) std::memcpy(out, s.c_str(), s.size() + 1);
    return s.size() + 1;
}
uint64_t sage_hip_hostdb_peptide_string(const SageHostDb* db, uint64_t i, char* out, uint64_t cap) {
    if (!db || i >= db->db.n_peptides()) return 0;
    return copy_out(db->db.peptide_string(i), out, cap);
}
uint64_t sage_hip_hostdb_peptide_proteins(const SageHostDb* db, uint64_t i, char* out, uint64_t cap) {
    if (!db || i >= db->db.n_peptides()) return 0;
    return copy_out(db->db.peptide_proteins(i), out, cap);
}
int sage_hip_hostdb_peptide_info(const SageHostDb* db, uint64_t i, uint32_t* num_proteins, uint8_t* semi_enzymatic) {
    if (!db || i >= db->db.n_peptides()) return fail(SAGE_HIP_ERR_INVALID, "peptide index out of range");
    if (num_proteins) *num_proteins = (uint32_t)(db->db.pep_protein_off[i + 1] - db->db.pep_protein_off[i]);
    if (semi_enzymatic) *semi_enzymatic = db->db.semi[i];
    return SAGE_HIP_OK;
}
struct SageMzml {
    MzmlRun run;
};
int sage_hip_mzml_read(const char* path, uint32_t file_id, int ms_level, SageMzml** out) {
    if (!path || !out) return fail(SAGE_HIP_ERR_INVALID, "sage_hip_mzml_read: null argument");
    auto h = std::make_unique<SageMzml>();
    std::string err;
    try {
        if (!read_mzml(path, file_id, ms_level, h->run, err)) return fail(SAGE_HIP_ERR_INVALID, err);
    } catch (const std::exception& e) {
        return fail(SAGE_HIP_ERR_INVALID, e.what());
    }
    *out = h.release();
    return SAGE_HIP_OK;
}
// Inputs the reference refuses to SEARCH (it panics while processing them; the reader itself accepts them): profile-mode MS2
// spectra (spectrum.rs:280-286 — Representation defaults to Profile, so a spectrum without the centroid term counts) and MS2
// spectra without a precursor (scoring.rs:466-468).  Call before preprocessing / scoring a run.
int sage_hip_mzml_check_searchable(const SageMzml* run) {
    if (!run) return fail(SAGE_HIP_ERR_INVALID, "sage_hip_mzml_check_searchable: null argument");
    const MzmlRun& r = run->run;
    for (uint64_t i = 0; i < r.n(); ++i) {
        if (r.ms_level[i] < 2) continue;
        const char* id = r.ids.data() + r.id_off[i];
        if (!r.centroid[i]) return fail(SAGE_HIP_ERR_INVALID, std::string("Scan ") + id + " contains profile data! Please convert to centroid");
        if (!r.has_precursor[i]) return fail(SAGE_HIP_ERR_INVALID, std::string("missing MS1 precursor for ") + id);
    }
    return SAGE_HIP_OK;
}
int sage_hip_mzml_view(const SageMzml* run, SageRawBatch* out) {
    if (!run || !out) return fail(SAGE_HIP_ERR_INVALID, "sage_hip_mzml_view: null argument");
    const MzmlRun& r = run->run;
    out->n_spectra = (uint32_t)r.n();
    out->peak_off = r.peak_off.data();
    out->mz = r.mz.data();
    out->intensities = r.intensities.data();
    out->precursor_mz = r.precursor_mz.data();
    out->precursor_charge = r.precursor_charge.data();
    out->isolation_lo = r.isolation_lo.data();
    out->isolation_hi = r.isolation_hi.data();
    out->scan_start_time = r.scan_start_time.data();
    out->inverse_ion_mobility = r.inverse_ion_mobility.data();
    out->file_id = r.file_id.data();
    return SAGE_HIP_OK;
}
const char* sage_hip_mzml_spectrum_id(const SageMzml* run, uint64_t i) {
    if (!run || i >= run->run.n()) return nullptr;
    return run->run.ids.data() + run->run.id_off[i];
}
void sage_hip_mzml_free(SageMzml* run) { delete run; }
int sage_hip_write_results(const char* path, int format, const SageHostDb* db, const SageFeature* features, uint64_t n,
                           const uint64_t* order, const uint64_t* psm_id, const char* const* filenames, uint32_t n_files,
                           const char* const* spec_ids, const SagePostColumns* post) {
    if (!path || !db || (n && (!features || !psm_id || !filenames || !spec_ids)))
        return fail(SAGE_HIP_ERR_INVALID, "sage_hip_write_results: null argument");
    if (format != SAGE_FORMAT_TSV && format != SAGE_FORMAT_PIN) return fail(SAGE_HIP_ERR_INVALID, "sage_hip_write_results: unknown format");
    for (uint64_t r = 0; order && r < n; ++r)
        if (order[r] >= n) return fail(SAGE_HIP_ERR_INVALID, "sage_hip_write_results: order entry out of range");
    std::string err;
    if (!write_results(path, format, db->db, features, n, order, psm_id, filenames, n_files, spec_ids, post, err))
        return fail(SAGE_HIP_ERR_INVALID, err);
    return SAGE_HIP_OK;
}
int sage_hip_hostdb_feature_peptides(const SageHostDb* db, const uint32_t* peptide_idx, uint64_t n, uint64_t* seq_off, uint8_t* seq,
                                     float* monoisotopic) {
    if (!db || (n && !peptide_idx) || !seq_off) return fail(SAGE_HIP_ERR_INVALID, "sage_hip_hostdb_feature_peptides: null argument");
    const HostDb& d = db->db;
    uint64_t off = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (peptide_idx[i] >= d.n_peptides()) return fail(SAGE_HIP_ERR_INVALID, "peptide index out of range");
        const uint64_t p = peptide_idx[i], len = d.seq_off[p + 1] - d.seq_off[p];
        seq_off[i] = off;
        if (seq) std::memcpy(seq + off, d.seq.data() + d.seq_off[p], len);
        if (monoisotopic) monoisotopic[i] = d.pep_mono[p];
        off += len;
    }
    seq_off[n] = off;
    return SAGE_HIP_OK;
}
int sage_hip_hostdb_competition_keys(const SageHostDb* db, const uint32_t* peptide_idx, uint64_t n, uint32_t* peptide_key,
                                     uint32_t* n_peptide_keys, uint32_t* protein_key, uint32_t* n_protein_keys) {
    if (!db || (n && (!peptide_idx || !peptide_key || !protein_key)) || !n_peptide_keys || !n_protein_keys)
        return fail(SAGE_HIP_ERR_INVALID, "sage_hip_hostdb_competition_keys: null argument");
    for (uint64_t i = 0; i < n; ++i)
        if (peptide_idx[i] >= db->db.n_peptides()) return fail(SAGE_HIP_ERR_INVALID, "peptide index out of range");
    db->db.competition_keys(peptide_idx, n, peptide_key, *n_peptide_keys, protein_key, *n_protein_keys);
    return SAGE_HIP_OK;
}
uint64_t sage_hip_process_ms2(uint64_t take_top_n, int deisotope, float min_deisotope_mz, const float* mz,
                              const float* intensity, uint64_t n, uint8_t precursor_charge, float* out_mass,
                              float* out_intensity, float* out_tic) {
    return process_ms2(take_top_n, deisotope != 0, min_deisotope_mz, mz, intensity, n, precursor_charge, out_mass,
                       out_intensity, out_tic);
}

// ---------------------------------------------------------------------------------------------
// device side
// ---------------------------------------------------------------------------------------------
int sage_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int sage_hip_db_create(const SageDbView* v, int device, SageDeviceDb** out) {
    if (!v || !out) return fail(SAGE_HIP_ERR_INVALID, "null argument");
    if (v->n_peptides >= 0xFFFFFFFFull) return fail(SAGE_HIP_ERR_UNSUPPORTED, "more than 2^32-2 peptides");
    if (v->n_ion_kinds > 8) return fail(SAGE_HIP_ERR_INVALID, "more than 8 ion kinds");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(SAGE_HIP_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(SAGE_HIP_ERR_INVALID, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    auto d = std::make_unique<SageDeviceDb>();
    d->device = device;
    const uint64_t np = v->n_peptides;
    const uint32_t nk = v->n_ion_kinds;

    // complete ion table for rescoring (IonSeries for every configured kind, ion_series.rs:36-85)
    std::vector<uint64_t> ion_off(np + 1, 0);
    std::vector<uint32_t> info(np);
    uint32_t max_ions = 0, max_len = 0;
    for (uint64_t i = 0; i < np; i++) {
        const uint64_t len = v->seq_off[i + 1] - v->seq_off[i];
        // (the production rescoring kernel keeps a candidate's longest-run state in 10-bit fields — kernels.hip, run_matched_packed —;
        // a database with longer peptides is scored by the general instances: scorer_init, DevScorer::long_runs)
        if (len > 65535) return fail(SAGE_HIP_ERR_UNSUPPORTED, "peptide longer than 65535 residues");  // (pep_info: 16 bits)
        max_len = std::max<uint32_t>(max_len, (uint32_t)len);
        const uint64_t cnt = (len ? len - 1 : 0) * nk;
        ion_off[i + 1] = ion_off[i] + cnt;
        max_ions = std::max<uint32_t>(max_ions, (uint32_t)cnt);
        info[i] = (uint32_t)len | ((uint32_t)(v->decoy[i] ? 1 : 0) << 16) | ((uint32_t)v->missed_cleavages[i] << 24);
    }
    uint32_t tile_shift = 15;
    if (const char* e = getenv("SAGE_HIP_TILE_SHIFT")) tile_shift = (uint32_t)std::min(15, std::max(11, atoi(e)));
    const uint64_t n_tiles = std::max<uint64_t>(1, (np + (1ull << tile_shift) - 1) >> tile_shift);
    // 1/256 Da cells.  The scale is a power of two, so `m/z * scale` is exact in f32 and a fragment-tolerance window
    // [lo, hi] maps to the cell range [floor(lo*scale), floor(hi*scale)] with no safety margin.
    const float lut_scale = 256.0f;
    uint32_t lut_stride = 0;
    uint64_t nf = v->n_fragments;
    std::vector<uint64_t> host_pm_off;  // fragment offset of every peptide (either branch)
    d->h_pep_mono.assign(v->pep_mono, v->pep_mono + np);
    if (!v->fragments) {
        // ---- Parameters::build_from_peptides (database.rs:265-346) on the device: index_build.hip ----
        std::vector<uint64_t> pm_off(np + 1, 0);
        for (uint64_t i = 0; i < np; i++) {
            const uint64_t len = v->seq_off[i + 1] - v->seq_off[i], lm1 = len ? len - 1 : 0;
            pm_off[i + 1] = pm_off[i] + (lm1 > v->min_ion_index ? lm1 - v->min_ion_index : 0) * nk;  // database.rs:281-292
        }
        nf = pm_off[np];
        if (nf >= 0xFFFFFFF0ull) return fail(SAGE_HIP_ERR_UNSUPPORTED, "more than 2^32-16 fragments");
        std::vector<uint64_t> tile_off(n_tiles + 1, 0);
        for (uint64_t t = 0; t < n_tiles; t++) tile_off[t + 1] = pm_off[std::min<uint64_t>(np, (t + 1) << tile_shift)];
        const uint64_t total_res = np ? v->seq_off[np] : 0;
        DevBuf<uint64_t> d_seq_off, d_tile_off;
        DevBuf<uint8_t> d_seq, d_kinds;
        DevBuf<float> d_mods, d_nterm;
        HIP_TRY(d_seq_off.upload(v->seq_off, np + 1));
        HIP_TRY(d_seq.upload(v->seq, total_res));
        HIP_TRY(d_mods.upload(v->mods, total_res));
        HIP_TRY(d_nterm.upload(v->nterm, np));
        HIP_TRY(d_kinds.upload(v->ion_kinds, nk));
        HIP_TRY(d_tile_off.upload(tile_off.data(), n_tiles + 1));
        HIP_TRY(d->pep_mono.upload(v->pep_mono, np));
        HIP_TRY(d->ion_off.upload(ion_off.data(), np + 1));
        HIP_TRY(d->pm_off.upload(pm_off.data(), np + 1));
        HIP_TRY(d->ions.alloc(ion_off[np] + 8));  // (padded: the rescoring kernel reads ions four at a time)
        HIP_TRY(d->pm_frag.alloc(nf));
        HIP_TRY(d->tm_frag.alloc(nf + 2));
        hipError_t be = (hipError_t)generate_fragments_on_device(np, nk, d_kinds.p, d_seq_off.p, d_seq.p, d_mods.p, d_nterm.p, d->pep_mono.p,
                                                                v->min_ion_index, d->ion_off.p, d->pm_off.p, d->ions.p, d->pm_frag.p, nullptr);
        uint32_t* lut_p = nullptr;
        if (be == hipSuccess)
            be = (hipError_t)build_tile_copy_on_device(d->pm_frag.p, nf, tile_shift, (uint32_t)n_tiles, d_tile_off.p, lut_scale,
                                                       d->tm_frag.p, &lut_p, &lut_stride, nullptr, TM_LUT_LAYOUT);
        if (be != hipSuccess)
            return fail(be == hipErrorOutOfMemory ? SAGE_HIP_ERR_OOM : SAGE_HIP_ERR_HIP, std::string("device index build: ") + hipGetErrorString(be));
        d->tm_lut.p = lut_p;
        d->tm_lut.n = (size_t)tm_lut_rows((uint32_t)n_tiles) * lut_stride;
        host_pm_off.swap(pm_off);
        HIP_TRY(d->pep_info.upload(info.data(), np));
    } else {
        // group IndexedDatabase.fragments by peptide (counting sort): tiles are runs of 2^tile_shift consecutive peptides
        std::vector<uint64_t> pm_off(np + 1, 0);
        for (uint64_t i = 0; i < nf; i++) {
            if (v->fragments[i].peptide_index >= np) return fail(SAGE_HIP_ERR_INVALID, "fragment peptide_index out of range");
            pm_off[v->fragments[i].peptide_index + 1]++;
        }
        for (uint64_t i = 0; i < np; i++) pm_off[i + 1] += pm_off[i];
        std::vector<SageTheoretical> tm(nf + 2, SageTheoretical{0xFFFFFFFFu, 0.0f});
        {
            std::vector<uint64_t> cur(pm_off.begin(), pm_off.end() - 1);
            for (uint64_t i = 0; i < nf; i++) tm[cur[v->fragments[i].peptide_index]++] = v->fragments[i];
        }
        HIP_TRY(d->pm_frag.upload(tm.data(), nf));  // (before the tiles are re-sorted by m/z below)
        HIP_TRY(d->pm_off.upload(pm_off.data(), np + 1));
        d->h_pep_mono.assign(v->pep_mono, v->pep_mono + np);
        std::vector<float> ions(ion_off[np] + 8);  // (padded: the rescoring kernel reads ions four at a time)
        parallel_for(np, 4096, [&](size_t ib, size_t ie, unsigned) {
            for (size_t i = ib; i < ie; i++) {
                const uint64_t len = v->seq_off[i + 1] - v->seq_off[i];
                const uint64_t lm1 = len ? len - 1 : 0;
                for (uint32_t k = 0; k < nk; k++)
                    ion_series_flat(v->seq + v->seq_off[i], v->mods + v->seq_off[i], len, v->nterm[i], v->pep_mono[i],
                                    v->ion_kinds[k], ions.data() + ion_off[i] + (uint64_t)k * lm1);
            }
        });
        // tile-major copy for large precursor windows: tile = peptide_index >> tile_shift, (m/z, peptide) order inside a
        // tile, and a per-tile position table tm_lut[t][c] = first position of tile t with m/z >= c / lut_scale.
        if (nf >= 0xFFFFFFF0ull) return fail(SAGE_HIP_ERR_UNSUPPORTED, "more than 2^32-16 fragments");
        std::vector<uint64_t> tile_off(n_tiles + 1, 0);
        for (uint64_t t = 0; t < n_tiles; t++) tile_off[t + 1] = pm_off[std::min<uint64_t>(np, (t + 1) << tile_shift)];
        parallel_for(n_tiles, 1, [&](size_t tb, size_t te, unsigned) {
            for (size_t t = tb; t < te; t++)
                std::sort(tm.begin() + tile_off[t], tm.begin() + tile_off[t + 1], [](const SageTheoretical& x, const SageTheoretical& y) {
                    const int32_t kx = sagecore::order_key(x.fragment_mz), ky = sagecore::order_key(y.fragment_mz);
                    return kx != ky ? kx < ky : x.peptide_index < y.peptide_index;
                });
        });
        float max_mz = 0.0f;
        for (uint64_t i = 0; i < nf; i++)
            if (tm[i].fragment_mz > max_mz && std::isfinite(tm[i].fragment_mz)) max_mz = tm[i].fragment_mz;
        lut_stride = (uint32_t)std::min<double>(std::ceil((double)max_mz * lut_scale) + 3.0, 64.0e6);
        if ((double)n_tiles * lut_stride > 4.0e9) return fail(SAGE_HIP_ERR_UNSUPPORTED, "tile position table larger than 16 GB");
        // (rows beyond the last tile — padding of the quad layout — are empty tiles at the end of the array)
        std::vector<uint32_t> lut((size_t)tm_lut_rows((uint32_t)n_tiles) * lut_stride, (uint32_t)tile_off[n_tiles]);
        parallel_for(n_tiles, 1, [&](size_t tb, size_t te, unsigned) {
            for (size_t t = tb; t < te; t++) {
                uint64_t pos = tile_off[t];
                const uint64_t tend = tile_off[t + 1];
                auto cell = [&](uint32_t c) -> uint32_t& { return lut[tm_lut_index((uint32_t)t, c, (uint32_t)n_tiles, lut_stride)]; };
                for (uint32_t c = 0; c < lut_stride; c++) {
                    const double edge = (double)c / (double)lut_scale;
                    // NaN and m/z beyond the table (non-finite or > 250 kDa) compare false and stay in the last cell's run
                    while (pos < tend && (double)tm[pos].fragment_mz < edge) pos++;
                    cell(c) = (uint32_t)pos;
                }
                cell(0) = (uint32_t)tile_off[t];  // a window starting below cell 0 starts at the tile's first entry
                cell(lut_stride - 1) = (uint32_t)tend;
            }
        });
        HIP_TRY(d->tm_frag.upload(tm.data(), tm.size()));
        HIP_TRY(d->tm_lut.upload(lut.data(), lut.size()));
        HIP_TRY(d->pep_mono.upload(v->pep_mono, np));
        HIP_TRY(d->ions.upload(ions.data(), ions.size()));
        HIP_TRY(d->ion_off.upload(ion_off.data(), np + 1));
        HIP_TRY(d->pep_info.upload(info.data(), np));
        host_pm_off.swap(pm_off);
    }
    {
        // small-tile copy for the narrow kernel (device_types.h: tm2_*), sorted on the device from the peptide-major list
        // 2^11 peptides per small tile (round 4; 2^12 before): a +-10 ppm window of a human digest holds ~200 candidates, so of a
        // tile's run for one fragment window only candidates / tile-size are inside the precursor window — the rest is fetched and
        // thrown away, and prelim_kernel follows its line traffic.  C3, prelim ms per 500 000 spectra by shift 13 / 12 / 11 / 10 / 9 / 8:
        // 2.26 / 1.86 / 1.68 / 1.67 / 1.83 / 2.20 (smaller tiles: more table rows, each reused by fewer spectra; 62 500-spectrum
        // steps favour 11 over 10).  The table doubles to 1.5 GB for C3.
        uint32_t tile2_shift = 11;
        if (const char* e = getenv("SAGE_HIP_TILE2_SHIFT")) tile2_shift = (uint32_t)std::min(16, std::max(6, atoi(e)));
        float lut2_scale = 32.0f;  // cells per Da (a power of two, like lut_scale)
        if (const char* e = getenv("SAGE_HIP_LUT2_SCALE")) {  // (experiments: 8 .. 256, rounded down to a power of two)
            const int v = std::min(256, std::max(8, atoi(e)));
            lut2_scale = (float)(1 << (31 - __builtin_clz((unsigned)v)));
        }
        const uint64_t n_tiles2 = std::max<uint64_t>(1, (np + (1ull << tile2_shift) - 1) >> tile2_shift);
        std::vector<uint64_t> tile2_off(n_tiles2 + 1, 0);
        for (uint64_t t = 0; t < n_tiles2; t++) tile2_off[t + 1] = host_pm_off[std::min<uint64_t>(np, (t + 1) << tile2_shift)];
        DevBuf<uint64_t> d_tile2_off;
        HIP_TRY(d_tile2_off.upload(tile2_off.data(), n_tiles2 + 1));
        HIP_TRY(d->tm2_frag.alloc(nf + 2));
        uint32_t* lut2_p = nullptr;
        uint32_t lut2_stride = 0;
        const hipError_t be = (hipError_t)build_tile_copy_on_device(d->pm_frag.p, nf, tile2_shift, (uint32_t)n_tiles2, d_tile2_off.p,
                                                                    lut2_scale, d->tm2_frag.p, &lut2_p, &lut2_stride, nullptr);
        if (be != hipSuccess)
            return fail(be == hipErrorOutOfMemory ? SAGE_HIP_ERR_OOM : SAGE_HIP_ERR_HIP, std::string("device index build: ") + hipGetErrorString(be));
        // the table in succinct form (index_build.hip: build_succinct_lut_on_device; the row-major table is its input only —
        // 1.5 GB for C3 that the narrow kernel no longer reads a line of per lookup)
        sagecore::LutWord* l1_p = nullptr;
        uint32_t* pos_p = nullptr;
        uint32_t lut2_words = 0;
        uint64_t n_pos = 0;
        const hipError_t se = (hipError_t)build_succinct_lut_on_device(lut2_p, (uint32_t)n_tiles2, lut2_stride, &l1_p, &pos_p, &lut2_words, &n_pos, nullptr);
        (void)hipFree(lut2_p);
        if (se != hipSuccess)
            return fail(se == hipErrorOutOfMemory ? SAGE_HIP_ERR_OOM : SAGE_HIP_ERR_HIP, std::string("device index build (succinct table): ") + hipGetErrorString(se));
        d->tm2_l1.p = l1_p;
        d->tm2_l1.n = (size_t)n_tiles2 * lut2_words;
        d->tm2_pos.p = pos_p;
        d->tm2_pos.n = (size_t)std::max<uint64_t>(n_pos, 1);
        d->view.tm2_frag = d->tm2_frag.p;
        d->view.tm2_l1 = d->tm2_l1.p;
        d->view.tm2_pos = d->tm2_pos.p;
        d->view.lut2_words = lut2_words;
        d->view.tile2_shift = tile2_shift;
        d->view.n_tiles2 = (uint32_t)n_tiles2;
        d->view.lut2_stride = lut2_stride;
        d->view.lut2_scale = lut2_scale;
    }
    d->max_ions = max_ions;
    d->max_len = max_len;
    HIP_TRY((hipError_t)ion_abs_range_on_device(d->ions.p, ion_off[np], &d->ion_lo_bits, &d->ion_hi_bits));
    {  // the position table of the precursor-window search key
        uint32_t* lut_p = nullptr;
        uint32_t bins = 0;
        float inv_w = 0.0f;
        // (SAGE_HIP_NO_PEP_LUT=1: without — every window through the full search; tests hold the two against each other)
        const bool off = getenv("SAGE_HIP_NO_PEP_LUT") && getenv("SAGE_HIP_NO_PEP_LUT")[0] == '1';
        const hipError_t be = off ? hipSuccess
                                  : (hipError_t)build_peptide_mass_lut(d->pep_mono.p, (uint32_t)np, np ? d->h_pep_mono[np - 1] : 0.0f, &lut_p, &bins, &inv_w, nullptr);
        if (be != hipSuccess)
            return fail(be == hipErrorOutOfMemory ? SAGE_HIP_ERR_OOM : SAGE_HIP_ERR_HIP, std::string("peptide-mass table: ") + hipGetErrorString(be));
        d->pep_lut.p = lut_p;
        d->pep_lut.n = lut_p ? (size_t)bins + 1 : 0;
        d->view.pep_lut = lut_p;
        d->view.pep_lut_bins = bins;
        d->view.pep_lut_inv_w = inv_w;
    }
    d->view.pep_mono = d->pep_mono.p;
    d->view.np = (uint32_t)np;
    d->view.pm_frag = d->pm_frag.p;
    d->view.pm_off = d->pm_off.p;
    d->view.ions = d->ions.p;
    d->view.ion_off = d->ion_off.p;
    d->view.pep_info = d->pep_info.p;
    d->view.tm_frag = d->tm_frag.p;
    d->view.tm_lut = d->tm_lut.p;
    d->view.tile_shift = tile_shift;
    d->view.n_tiles = (uint32_t)n_tiles;
    d->view.lut_stride = lut_stride;
    d->view.lut_scale = lut_scale;
    d->view.nf = nf;
    std::memset(d->view.ion_kinds, 0, sizeof d->view.ion_kinds);
    for (uint32_t k = 0; k < nk; k++) d->view.ion_kinds[k] = v->ion_kinds[k];
    d->view.n_kinds = nk;
    if (!getenv("SAGE_HIP_KEEP_PM_FRAG")) {
        d->pm_frag.release();
        d->view.pm_frag = nullptr;
    }
    d->bytes = d->pep_mono.bytes() + d->pep_lut.bytes() + d->pm_frag.bytes() + d->pm_off.bytes() + d->ions.bytes() + d->ion_off.bytes() +
               d->pep_info.bytes() + d->tm_frag.bytes() + d->tm_lut.bytes() + d->tm2_frag.bytes() + d->tm2_l1.bytes() + d->tm2_pos.bytes();
    *out = d.release();
    return SAGE_HIP_OK;
}

// the peptide-major fragment list for a batch that takes the stream variant of the preliminary kernels (SageDeviceDb::pm_frag)
static int ensure_pm_frag(SageDeviceDb* db) {
    std::lock_guard<std::mutex> lock(db->pm_mu);
    if (db->view.pm_frag) return SAGE_HIP_OK;
    HIP_TRY(db->pm_frag.alloc((size_t)db->view.nf + 2));
    const hipError_t be = (hipError_t)rebuild_peptide_major_on_device(db->tm_frag.p, db->view.nf, db->pm_frag.p, nullptr);
    if (be != hipSuccess) {
        db->pm_frag.release();
        return fail(be == hipErrorOutOfMemory ? SAGE_HIP_ERR_OOM : SAGE_HIP_ERR_HIP, std::string("peptide-major fragment list: ") + hipGetErrorString(be));
    }
    db->bytes += db->pm_frag.bytes();
    db->view.pm_frag = db->pm_frag.p;
    return SAGE_HIP_OK;
}

void sage_hip_db_destroy(SageDeviceDb* db) {
    if (!db) return;
    (void)hipSetDevice(db->device);
    delete db;
}
uint64_t sage_hip_db_device_bytes(const SageDeviceDb* db) { return db ? db->bytes : 0; }

// DevScorer::tol_mode (core.h: TolMode).  TOL_SYM: a symmetric ppm fragment tolerance — one division per Tolerance::bounds
// (mass.rs:21-35).  TOL_FAST: the rescoring instance with the short divisions by 1e6 and by 3 (core.h: div_const_fast —
// the correctly rounded quotient for every dividend of FAST_DIV_LO <= |x| <= FAST_DIV_HI) may score for this scorer: the
// dividends are bounded from the database's ion table — an ion for the charge-3 m/z (scoring.rs:707), |ion| / charge (a u8:
// <= 255) times each bound of a ppm tolerance — and whatever does not fit keeps the IEEE sequence: a table with a zero, an
// infinite or a NaN entry, a ppm bound of zero, absurd magnitudes.
static uint32_t scorer_tol_mode(const sagecore::Tol& t, uint32_t ion_lo_bits, uint32_t ion_hi_bits) {
    uint32_t mode = 0;
    if (t.kind == 0 && t.lo == -t.hi) mode |= sagecore::TOL_SYM;
    if (ion_lo_bits > ion_hi_bits) return mode;  // (no ions at all)
    float ion_lo, ion_hi;
    std::memcpy(&ion_lo, &ion_lo_bits, 4);
    std::memcpy(&ion_hi, &ion_hi_bits, 4);
    const double lo = sagecore::FAST_DIV_LO, hi = sagecore::FAST_DIV_HI;
    bool ok = ion_lo >= lo && ion_hi <= hi;  // (NaN: its bits sort above +inf, the comparison fails)
    if (ok && t.kind == 0)
        for (const float b : {t.lo, t.hi}) {
            const double a = std::fabs((double)b);
            // (a factor of two either side for the rounding of the f32 quotient |ion| / charge and of the f32 product)
            ok = ok && a > 0.0 && (double)ion_lo / 255.0 * a >= 2.0 * lo && (double)ion_hi * a <= 0.5 * hi;
        }
    if (ok) mode |= sagecore::TOL_FAST;
    return mode;
}

static int scorer_init(SageScorer* sp, SageDeviceDb* db, const SageScorerParams* p) {
    SageScorer* s = sp;
    s->db = db;
    s->params = *p;
    DevScorer& d = s->dev;
    d.precursor_tol = {p->precursor_tol.kind, p->precursor_tol.lo, p->precursor_tol.hi};
    d.fragment_tol = {p->fragment_tol.kind, p->fragment_tol.lo, p->fragment_tol.hi};
    d.pbm_reach = sagecore::pbm_reach_of(d.fragment_tol);
    d.min_matched_peaks = p->min_matched_peaks;
    d.min_isotope_err = p->min_isotope_err;
    d.max_isotope_err = p->max_isotope_err;
    d.min_precursor_charge = p->min_precursor_charge;
    d.max_precursor_charge = p->max_precursor_charge;
    d.override_precursor_charge = p->override_precursor_charge;
    d.max_fragment_charge = p->max_fragment_charge;
    d.chimera = p->chimera;
    d.report_psms = p->report_psms;
    d.wide_window = p->wide_window;
    d.score_type = p->score_type;
    d.kmax = std::max<uint32_t>(50, 2 * p->report_psms);
    const uint32_t n_iso = (uint32_t)(p->max_isotope_err - p->min_isotope_err) + 1;
    const uint32_t n_z = (uint32_t)(p->max_precursor_charge - p->min_precursor_charge) + 1;
    d.list_cap = d.kmax * (std::max(n_iso, n_z) + 1);
    d.wcap = 1024;
    d.dbg_flags = 0;
    if (const char* e = getenv("SAGE_HIP_DEBUG_FLAGS")) d.dbg_flags = (uint32_t)atoi(e);
    d.exact = 0;
    d.xcd_chunk = 1024;
    if (const char* e = getenv("SAGE_HIP_XCD_CHUNK")) d.xcd_chunk = (uint32_t)std::max(0, atoi(e));
    // (bit 31: heaviest precursors first — the default since round 6: the expensive spectra start first and the step's tail is made
    // of cheap ones; a rank's 62 500-spectrum C3 shard -7 %, C2 -4 %, 500 000 spectra -1 %: scripts/experiments/r06_lab/gpu_r7y.sh)
    // Batches that may hold large windows keep the ascending order unless the variable asks (C4 +1 % with it, C5 -0.5 %: their
    // steps are the count kernel's, which takes the spectra in queue order): enqueue_compute.
    {
        const char* e = getenv("SAGE_HIP_SCHED_DESC");
        if (!e || atoi(e)) d.xcd_chunk |= 0x80000000u;
        s->sched_desc_forced = e != nullptr;
    }
    if (const char* e = getenv("SAGE_HIP_EXACT")) s->exact_always = atoi(e) != 0;
    if (const char* e = getenv("SAGE_HIP_FUSED")) s->fused = atoi(e) != 0;
    if (const char* e = getenv("SAGE_HIP_ONE_LANE")) s->two_lanes = atoi(e) == 0;
    if (const char* e = getenv("SAGE_HIP_SEARCH_LAG")) s->search_lag = (uint32_t)std::max(0, atoi(e));
    if (const char* e = getenv("SAGE_HIP_REPLAY_WAVE_MAX")) s->replay_split = (uint64_t)atoll(e) & 0xFFFFFFFFull;
    if (const char* e = getenv("SAGE_HIP_REPLAY_LANE_MAX")) s->replay_split |= (uint64_t)(uint32_t)atoll(e) << 32;
    if (const char* e = getenv("SAGE_HIP_ONE_LAUNCH")) s->one_launch = atoi(e) != 0;
#ifndef SAGE_HIP_EXPERIMENTS
    // the losing first-pass experiments of DESIGN.md 4.7 (search_kernel, the fused kernel as the first pass) are compiled into
    // builds with -DSAGE_HIP_EXPERIMENTS only (scripts/variants.sh exp:"-DSAGE_HIP_EXPERIMENTS")
    if (s->one_launch || s->fused || getenv("SAGE_HIP_SEARCH_LAG"))
        return fail(SAGE_HIP_ERR_UNSUPPORTED, "SAGE_HIP_ONE_LAUNCH / SAGE_HIP_FUSED / SAGE_HIP_SEARCH_LAG need a build with -DSAGE_HIP_EXPERIMENTS");
#endif
    if (const char* e = getenv("SAGE_HIP_NO_ZEROCOPY")) s->zero_copy = atoi(e) == 0;
    if (const char* e = getenv("SAGE_HIP_NO_FAST_TIES")) s->fast_ties = atoi(e) == 0;
    if (const char* e = getenv("SAGE_HIP_WCAP")) d.wcap = (uint32_t)std::min(16384, std::max(64, atoi(e)));  // (32-bit heap keys need <= 65536)
    if (const char* e = getenv("SAGE_HIP_CHUNK")) s->chunk = (uint32_t)std::min(1 << 22, std::max(64, atoi(e)));
    HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&s->up_stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&s->down_stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&s->side_stream, hipStreamNonBlocking));
    if (const char* e = getenv("SAGE_HIP_WAYS")) s->ways = (uint32_t)std::min(4, std::max(1, atoi(e)));
    // cheap ties: only where the tied candidates' records are final whichever wins (one reported PSM, no chimera rounds)
    if (p->report_psms != 1 || p->chimera) s->fast_ties = false;
    // ... and only where a spectrum's window counts fit the LDS rescore_kernel replays them in (the peak bitmap + the peak table,
    // dead by then: kernels.hip FAST_TIE_WORDS; wcap <= 2560 — the default is 1024, SAGE_HIP_WCAP raises it).  Larger windows
    // settle their ties through the exact retry pass.
    if ((d.wcap + 1) / 2 > fast_tie_lds_words()) s->fast_ties = false;
    s->cnt_stride = 4u + (((d.wcap + 1) / 2 + 3u) & ~3u);  // (kernels.hip: CNT_ROW_HEADER)
    // Peptides of more than 1023 residues (the reference has no limit: database.rs:96-115 `max_len` is the user's): their ion
    // indices do not fit the one-register Run state of the production rescoring kernel, so such a database takes the same slow,
    // general instances as lists wider than a wavefront, with rescore_big_kernel's two-register Run form.
    d.long_runs = db->max_len > 1023 ? 1u : 0u;
    d.big_path = (d.kmax > 64 || d.long_runs) ? 1u : 0u;
    d.tol_mode = scorer_tol_mode(d.fragment_tol, db->ion_lo_bits, db->ion_hi_bits);
    // (the instance with the short divisions is opt-in: 97 vector instructions and 13 division sequences shorter on paper, identical
    // results — and 1-2 % SLOWER on C3, 2.755 against 2.70-2.73 ms per 500 000 spectra: profiles/r05_C3_short_divisions.txt)
    if (!getenv("SAGE_HIP_SHORT_DIVISIONS")) d.tol_mode &= ~(uint32_t)sagecore::TOL_FAST;
    if (d.big_path) {
        // report_psms > 32: preliminary lists of up to 256 candidates, wider than a wavefront.  The BIGK kernels (kernels.hip): heaps
        // in LDS, every trim exact (no order-free trims, hence no retry pass), rescoring 64 candidates at a time.
        s->kstride = ((d.kmax + 63) / 64) * 64;
        HIP_TRY((hipError_t)bigk_kernel_prepare(160 * 1024));
        s->exact_always = true;
        s->fused = s->one_launch = false;
        s->ways = 1;
    }
    for (hipStream_t& ws : s->way_stream) HIP_TRY(hipStreamCreateWithFlags(&ws, hipStreamNonBlocking));
    HIP_TRY(s->way_fork.create(false));
    for (Event& e : s->way_join) HIP_TRY(e.create(false));
    HIP_TRY(s->way_begin.create(true));
    for (Event& e : s->way_end) HIP_TRY(e.create(true));
    HIP_TRY(s->side_fork.create(false));
    HIP_TRY(s->side_join.create(false));
    for (OutSet& o : s->outs) {
        for (auto& e : o.ev) HIP_TRY(e.create(true));
        HIP_TRY(o.comp_done.create(false));
        HIP_TRY(o.down_done.create(false));
        HIP_TRY(o.counters.alloc(2 * CTR_COUNT));
        HIP_TRY(hipHostMalloc((void**)&o.h_counters, 2 * CTR_COUNT * 4, hipHostMallocDefault));
        HIP_TRY(hipHostGetDevicePointer((void**)&o.h_counters_view, o.h_counters, 0));
    }
    for (SageDeviceBatch& b : s->slots) {
        b.device = db->device;
        HIP_TRY(b.up_done.create(false));
    }
    // lnfact (scoring.rs:170-177) tabulated on the host with the SAME correctly rounded ln the kernels use (crlog.h): the
    // factorial terms do not depend on the host's libm
    std::vector<double> tbl(131072);  // every u16 count and every sum of two (Score.matched_b/y are u16, scoring.rs:17-30)
    tbl[0] = 1.0;
    for (uint32_t n = 1; n < tbl.size(); n++) {
        const double x = (double)n;
        tbl[n] = x * sagecore::cr_log(x) - x + 0.5 * sagecore::cr_log(x) + 0.5 * sagecore::cr_log(M_PI * 2.0 * x);
    }
    HIP_TRY(s->lnfact.upload(tbl.data(), tbl.size()));
    // large-window kernel: persistent workgroups, as many as the LDS tiles allow to be resident
    {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, db->device));
        const size_t tile_lds = ((size_t)1 << db->view.tile_shift) * 2 + 12 * 1024;  // counters + bitmap + windows (typical)
        const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / tile_lds));
        s->tile_blocks = (uint32_t)prop.multiProcessorCount * per_cu;
        const size_t tile_lds8 = ((size_t)1 << db->view.tile_shift) + 14 * 1024;
        s->tile_blocks8 = (uint32_t)prop.multiProcessorCount * (uint32_t)std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / tile_lds8));
        if (const char* e = getenv("SAGE_HIP_TILE_BLOCKS")) s->tile_blocks = s->tile_blocks8 = (uint32_t)std::max(1, atoi(e));
        if (const char* e = getenv("SAGE_HIP_NO_U8")) s->cnt8 = atoi(e) == 0;
        if (const char* e = getenv("SAGE_HIP_NO_REUSE")) s->reuse_counts = atoi(e) == 0;
        HIP_TRY((hipError_t)tile_kernel_prepare(160 * 1024));
        HIP_TRY((hipError_t)spectrum_kernel_prepare(160 * 1024));
    }
    s->qmax = queries_per_spectrum(d);
    if (const char* e = getenv("SAGE_HIP_PHASE_CLOCKS")) {
        if (atoi(e) > 0) {
            HIP_TRY(s->dbg.alloc(4096 * 32));
            HIP_TRY(hipMemset(s->dbg.p, 0, 4096 * 32 * 8));
        }
    }
    return SAGE_HIP_OK;
}

static void scorer_release(SageScorer* s) {
    (void)hipSetDevice(s->db->device);
    for (hipStream_t st : {s->stream, s->up_stream, s->down_stream, s->side_stream, s->way_stream[0], s->way_stream[1], s->way_stream[2]})
        if (st) {
            (void)hipStreamSynchronize(st);
            (void)hipStreamDestroy(st);
        }
    delete s;
}

int sage_hip_scorer_create(SageDeviceDb* db, const SageScorerParams* p, SageScorer** out) {
    if (!db || !p || !out) return fail(SAGE_HIP_ERR_INVALID, "null argument");
    if (p->report_psms == 0) return fail(SAGE_HIP_ERR_INVALID, "report_psms must be >= 1");
    // (trim_hits keeps max(50, 2 * report_psms) candidates, scoring.rs:322-329, and the reference has no cap; lists that no longer fit
    // a compute unit's LDS live in a global-memory workspace — enqueue_compute: hugebuf.  16-bit fields of the large-window
    // records bound the list at 65 534 entries.)
    if (p->report_psms > 32767) return fail(SAGE_HIP_ERR_UNSUPPORTED, "report_psms > 32767");
    if (p->min_isotope_err > p->max_isotope_err) return fail(SAGE_HIP_ERR_INVALID, "min_isotope_err > max_isotope_err");
    if (p->min_precursor_charge > p->max_precursor_charge || p->min_precursor_charge == 0)
        return fail(SAGE_HIP_ERR_INVALID, "precursor charge range must be [lo >= 1, hi >= lo]");
    if (p->score_type != 0 && p->score_type != 1) return fail(SAGE_HIP_ERR_INVALID, "unknown score_type");
    HIP_TRY(hipSetDevice(db->device));
    SageScorer* s = new SageScorer();
    const int rc = scorer_init(s, db, p);
    if (rc != SAGE_HIP_OK) {
        s->db = db;
        scorer_release(s);
        return rc;
    }
    *out = s;
    return SAGE_HIP_OK;
}

// Scorer::score takes &self and is called from every rayon worker at once (scoring.rs:300, runner.rs:311-325).  One handle
// may be shared between host threads — its calls then run one after the other — and a clone is a second handle on the same
// device database with its own streams and working set, for callers that want their batches scored concurrently.
int sage_hip_scorer_clone(SageScorer* scorer, SageScorer** out) {
    if (!scorer || !out) return fail(SAGE_HIP_ERR_INVALID, "null argument");
    return sage_hip_scorer_create(scorer->db, &scorer->params, out);
}

void sage_hip_scorer_destroy(SageScorer* s) {
    if (!s) return;
    scorer_release(s);
}

// ---- batches ---------------------------------------------------------------------------------------------------------------
// the device-side address of page-locked, mapped host memory (sage_hip_host_alloc / hipHostMalloc); null for anything else
static void* device_view(const void* p) {
    if (!p) return nullptr;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return at.type == hipMemoryTypeHost ? at.devicePointer : nullptr;
}
// ... without asking the driver, for memory this library page-locked itself (sage_hip_host_alloc keeps a registry of its blocks: a
// caller that scores batch after batch into the same result arrays — every step of a resident loop — otherwise pays two driver
// queries, ~5 us each, in front of the step's first launch).  Only those blocks: for page-locked memory of any other origin the
// library cannot know when it is unpinned or freed, so it asks every time.
struct HostBlock {
    const unsigned char* host;
    unsigned char* dev;
    size_t bytes;
};
static std::mutex g_blocks_mu;
static std::vector<HostBlock> g_blocks;
static void* device_view_cached(const void* p) {
    if (!p) return nullptr;
    {
        std::lock_guard<std::mutex> lock(g_blocks_mu);
        for (const HostBlock& b : g_blocks)
            if ((const unsigned char*)p >= b.host && (const unsigned char*)p < b.host + b.bytes) return b.dev + ((const unsigned char*)p - b.host);
    }
    return device_view(p);
}
static bool is_page_locked(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();  // plain pageable memory: "invalid value", not an error for us
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

// What a batch looks like to the narrow kernel, from (a sample of) its spectra: the mean candidate-window size picks the
// matching variant, the largest one says whether the large-window kernels have to be launched at all.
struct WindowEstimate {
    uint32_t probe;       // narrow kernel variant (DevBatchView::probe)
    bool maybe_wide;      // some window may exceed DevScorer::wcap (a sample cannot rule it out: callers re-run on a wrong "no")
};
static WindowEstimate choose_probe(const SageScorer* s, uint32_t n, const float* precursor_mz, const uint8_t* precursor_charge,
                                   const float* isolation_lo, const float* isolation_hi) {
    const SageScorerParams& p = s->params;
    const std::vector<float>& pm = s->db->h_pep_mono;
    const uint32_t step = std::max<uint32_t>(1, n / 1024);  // (each sample costs two searches over the whole peptide list: ~0.2 us)
    double sum = 0.0;
    uint64_t widest = 0;
    uint32_t cnt = 0;
    for (uint32_t i = 0; i < n; i += step, cnt++) {
        const bool ranged = p.wide_window || precursor_charge[i] == 0 || p.override_precursor_charge;  // scoring.rs:423, 437, 442
        const uint32_t z0 = ranged ? p.min_precursor_charge : precursor_charge[i], z1 = ranged ? p.max_precursor_charge : precursor_charge[i];
        for (uint32_t z = z0; z <= z1; z++) {
            const float center = (precursor_mz[i] - sagecore::PROTON) * (float)z;
            sagecore::Tol tol{p.precursor_tol.kind, p.precursor_tol.lo, p.precursor_tol.hi};
            if (p.wide_window) {
                float lo = -2.4f, hi = 2.4f;
                if (isolation_lo && isolation_hi && isolation_lo[i] == isolation_lo[i] && isolation_hi[i] == isolation_hi[i]) {
                    lo = isolation_lo[i];
                    hi = isolation_hi[i];
                }
                tol = sagecore::Tol{2, lo * (float)z, hi * (float)z};
            }
            float lo, hi;
            sagecore::tol_bounds(tol, center, lo, hi);
            // the window's upper end is searched for from its lower end, in doubling strides (windows are short next to the list)
            const auto first = std::lower_bound(pm.begin(), pm.end(), lo);
            size_t reach = 64;
            while ((size_t)(pm.end() - first) > reach && first[reach] <= hi) reach *= 2;
            const auto last = std::upper_bound(first, (size_t)(pm.end() - first) > reach ? first + reach : pm.end(), hi);
            const uint64_t wdw = (uint64_t)(last - first);
            if (z == z0) sum += (double)wdw;
            widest = std::max(widest, wdw);
        }
    }
    const double mean_window = cnt ? sum / cnt : 0.0;
    WindowEstimate e;
    // The matching variant: table lookups per (peak, fragment charge) window ("probe") unless the windows hold next to nothing.
    // Round 5 re-measured the threshold (round 2 had put it at 96 candidates, before the small tiles, the position table of the
    // peptide masses and the DPP scans made the probe variant 2-3x faster): C2, mean window ~25 candidates — preliminary kernel 0.82 ms
    // streaming against 0.33 ms probing per 50 000 spectra (the step 0.70 -> 0.50 ms); the heaviest 1/32 of C3 (mean window 67
    // candidates of ~80 fragments each) 0.79 against 0.12 ms per 15 600 spectra (profiles/r05_shard_sizes.txt).  Streaming walks
    // candidates x fragments entries with a binary search over the windows each; it can only win where that product is tiny.
    e.probe = mean_window > 4.0 ? 1u : 0u;
    if (const char* v = getenv("SAGE_HIP_NARROW")) e.probe = std::string(v) == "probe" ? 1u : std::string(v) == "stream" ? 0u : e.probe;
    // (a quarter of headroom between the widest sampled window and the capacity: isotope errors shift the centre, the sample is thin)
    e.maybe_wide = widest + widest / 4 + 2 > s->dev.wcap;
    if (const char* v = getenv("SAGE_HIP_ASSUME_NARROW")) e.maybe_wide = atoi(v) == 0;  // (tests: force the wrong guess and its repair)
    return e;
}

// largest fragment charge any spectrum of a batch can ask for (scoring.rs:239-247)
static uint32_t batch_fzcap(const SageScorerParams& p, uint32_t zmax, bool any_unknown) {
    uint32_t fzcap = 1;
    const bool ranged = p.wide_window || p.override_precursor_charge || any_unknown;
    for (uint32_t z = 1; z <= 255; z++) {
        const bool used = (ranged && z >= p.min_precursor_charge && z <= p.max_precursor_charge) ||
                          (!p.wide_window && !p.override_precursor_charge && z <= zmax);
        if (used) fzcap = std::max(fzcap, sagecore::max_fragment_charge(p.max_fragment_charge, z) - 1);
    }
    return fzcap;
}

// Score.matched_b / matched_y are u16 in the reference (scoring.rs:21-22) and the rescoring kernel keeps the pair packed in one
// register (core.h: the `mm` counter): a peptide / charge combination that could match more than 65535 ions of one side would
// carry from one half into the other where the reference's release build wraps a u16 (and its debug build panics).  Refuse it:
// n-terminal resp. c-terminal ion kinds x ions per kind x fragment charges of the batch.
static int check_match_counters(const SageScorer* s, uint32_t fzcap) {
    const DevDbView& db = s->db->view;
    uint32_t nterm = 0;
    for (uint32_t k = 0; k < db.n_kinds; k++) nterm += db.ion_kinds[k] <= 2 ? 1u : 0u;
    const uint64_t per_kind = db.n_kinds ? s->db->max_ions / db.n_kinds : 0;  // (max_ions: ions of the longest peptide, all kinds)
    const uint64_t worst = std::max<uint64_t>(nterm, db.n_kinds - nterm) * per_kind * fzcap;
    if (worst > 65535)
        return fail(SAGE_HIP_ERR_UNSUPPORTED, "a candidate could match " + std::to_string(worst) + " ions of one terminus (" +
                                                  std::to_string(per_kind) + " ions per kind x fragment charges up to " + std::to_string(fzcap) +
                                                  "): beyond the u16 match counters of Score (scoring.rs:21-22)");
    return SAGE_HIP_OK;
}

// Spectra [c0, c1) of a host batch -> the device arrays of `d`, asynchronously on `up`:
//   * the per-spectrum arrays (40 bytes per spectrum) are staged through d's page-locked block (peak offsets rebased to the
//     range, the launch limits pcap / fzcap found on the way);
//   * the peak arrays go by DMA straight from the caller's memory when it is page-locked (sage_hip_host_alloc), else through
//     the staging block (copied by a few host threads).
// The caller must have waited for d->up_done before (the staging block is being rewritten).  `probe`: narrow-kernel variant.
static int stage_and_upload(SageScorer* s, SageDeviceBatch* d, const SageSpectrumBatch* b, uint32_t c0, uint32_t c1, bool peaks_locked,
                            const WindowEstimate& est, hipStream_t up) {
    const uint32_t probe = est.probe;
    d->maybe_wide = est.maybe_wide;
    const uint32_t n = c1 - c0;
    const uint64_t base = n ? b->peak_off[c0] : 0, total = n ? b->peak_off[c1] - base : 0;
    if (n && b->peak_off[c1] < base) return fail(SAGE_HIP_ERR_INVALID, "peak_off is not monotone");
    const bool has_iso = b->isolation_lo && b->isolation_hi, has_rt = b->scan_start_time != nullptr,
               has_ims = b->inverse_ion_mobility != nullptr, has_fid = b->file_id != nullptr;
    d->n = n;
    HIP_TRY(d->masses.reserve(total));
    HIP_TRY(d->intensities.reserve(total));
    HIP_TRY(d->order.reserve(n));
    HIP_TRY(d->sort_a.reserve(n));
    HIP_TRY(d->sort_b.reserve(n));
    HIP_TRY(d->sort_idx.reserve(n));
    const size_t sort_bytes = schedule_temp_bytes(n);
    HIP_TRY(d->sort_tmp.reserve(sort_bytes));
    const bool use_sched = !getenv("SAGE_HIP_NO_SCHED");
    if (use_sched) { HIP_TRY(d->sched.reserve(2 * (size_t)n)); }
    const size_t small = ((size_t)n + 1) * 8 + (size_t)n * (4 * 7 + 1) + 64 * 12;
    HIP_TRY(d->stage.reserve(small + (peaks_locked ? 0 : total * 8 + 128)));
    HIP_TRY(d->meta.reserve(small));
    unsigned char* cur = d->stage.p;
    uint64_t* h_off = carve<uint64_t>(cur, (size_t)n + 1);
    float* h_mz = carve<float>(cur, n);
    uint8_t* h_z = carve<uint8_t>(cur, n);
    float* h_tic = carve<float>(cur, n);
    float* h_lo = has_iso ? carve<float>(cur, n) : nullptr;
    float* h_hi = has_iso ? carve<float>(cur, n) : nullptr;
    float* h_rt = has_rt ? carve<float>(cur, n) : nullptr;
    float* h_ims = has_ims ? carve<float>(cur, n) : nullptr;
    uint32_t* h_fid = has_fid ? carve<uint32_t>(cur, n) : nullptr;
    uint32_t pcap = 1, zmax = 0;
    bool any_unknown = false;
    h_off[0] = 0;
    // page-locked peak arrays go by DMA from where they lie: enqueued FIRST, so that the link is busy while this thread stages
    // the per-spectrum block below (0.3 ms per 131 072 spectra — at the head of a call nothing else hides it)
    const bool peaks_first = peaks_locked && n && total && b->masses && b->intensities;
    if (peaks_first) {
        HIP_TRY(hipMemcpyAsync(d->masses.p, b->masses + base, total * 4, hipMemcpyHostToDevice, up));
        HIP_TRY(hipMemcpyAsync(d->intensities.p, b->intensities + base, total * 4, hipMemcpyHostToDevice, up));
    }
    for (uint32_t i = 0; i < n; i++) {
        const uint64_t a = b->peak_off[c0 + i], e = b->peak_off[c0 + i + 1];
        if (e < a) return fail(SAGE_HIP_ERR_INVALID, "peak_off is not monotone");
        h_off[i + 1] = e - base;
        pcap = std::max<uint32_t>(pcap, (uint32_t)std::min<uint64_t>(e - a, 0xFFFFFFFFull));
        const uint8_t z = b->precursor_charge[c0 + i];
        zmax = std::max<uint32_t>(zmax, z);
        any_unknown = any_unknown || z == 0;
    }
    if (n) {
        std::memcpy(h_mz, b->precursor_mz + c0, (size_t)n * 4);
        std::memcpy(h_z, b->precursor_charge + c0, n);
        std::memcpy(h_tic, b->total_ion_current + c0, (size_t)n * 4);
        if (has_iso) {
            std::memcpy(h_lo, b->isolation_lo + c0, (size_t)n * 4);
            std::memcpy(h_hi, b->isolation_hi + c0, (size_t)n * 4);
        }
        if (has_rt) std::memcpy(h_rt, b->scan_start_time + c0, (size_t)n * 4);
        if (has_ims) std::memcpy(h_ims, b->inverse_ion_mobility + c0, (size_t)n * 4);
        if (has_fid) std::memcpy(h_fid, b->file_id + c0, (size_t)n * 4);
    }
    const size_t meta_bytes = (size_t)(cur - d->stage.p);
    auto image = [&](const void* h) { return h ? d->meta.p + ((const unsigned char*)h - d->stage.p) : nullptr; };
    const float* src_m = b->masses ? b->masses + base : nullptr;
    const float* src_i = b->intensities ? b->intensities + base : nullptr;
    if (!peaks_locked && total) {
        float* h_m = carve<float>(cur, total);
        float* h_i = carve<float>(cur, total);
        const float *pm_ = src_m, *pi_ = src_i;
        parallel_for(total, 1u << 20, [&](size_t ib, size_t ie, unsigned) {
            std::memcpy(h_m + ib, pm_ + ib, (ie - ib) * 4);
            std::memcpy(h_i + ib, pi_ + ib, (ie - ib) * 4);
        });
        src_m = h_m;
        src_i = h_i;
    }
    if (n) {
        // the per-spectrum arrays: one copy of the staging block's head, the device pointers are its image
        // ... and the launch schedule (ascending neutral precursor mass) is sorted from it on the device, both on a stream of their
        // own WHILE the peaks cross the link on `up` (behind the uploads, on their stream, the sort's dozen small kernels left the
        // link idle for ~0.1 ms per chunk of the streaming pipeline).  (The previous users of this slot's buffers are done: every
        // caller has waited for the slot's last batch.)
        HIP_TRY(d->sort_done.create(false));
        HIP_TRY(hipMemcpyAsync(d->meta.p, d->stage.p, meta_bytes, hipMemcpyHostToDevice, s->side_stream));
        HIP_TRY((hipError_t)schedule_on_device(n, (const float*)image(h_mz), (const uint8_t*)image(h_z), s->params.min_precursor_charge,
                                               d->sort_a.p, d->sort_b.p, d->sort_idx.p, d->order.p, d->sort_tmp.p, sort_bytes, s->side_stream));
        if (use_sched)
            schedule_records_on_device(n, d->order.p, (const uint64_t*)image(h_off), (const float*)image(h_mz), (const uint8_t*)image(h_z),
                                       (const float*)image(h_lo), (const float*)image(h_hi), d->sched.p, s->side_stream);
        HIP_TRY(hipEventRecord(d->sort_done.e, s->side_stream));
        if (total && !peaks_first) {
            HIP_TRY(hipMemcpyAsync(d->masses.p, src_m, total * 4, hipMemcpyHostToDevice, up));
            HIP_TRY(hipMemcpyAsync(d->intensities.p, src_i, total * 4, hipMemcpyHostToDevice, up));
        }
        HIP_TRY(hipStreamWaitEvent(up, d->sort_done.e, 0));  // (`up_done` below stands for both)
    }
    HIP_TRY(hipEventRecord(d->up_done.e, up));
    DevBatchView& v = d->view;
    v = DevBatchView{};
    v.n = n;
    v.n_dev = nullptr;
    v.spec_base = c0;
    v.peak_off = (const uint64_t*)image(h_off);
    v.masses = d->masses.p;
    v.intensities = d->intensities.p;
    v.precursor_mz = (const float*)image(h_mz);
    v.precursor_charge = (const uint8_t*)image(h_z);
    v.isolation_lo = (const float*)image(h_lo);
    v.isolation_hi = (const float*)image(h_hi);
    v.tic = (const float*)image(h_tic);
    v.rt = (const float*)image(h_rt);
    v.ims = (const float*)image(h_ims);
    v.file_id = (const uint32_t*)image(h_fid);
    v.order = d->order.p;
    v.sched = use_sched && n ? d->sched.p : nullptr;
    v.probe = probe;
    v.pcap = pcap;
    v.fzcap = batch_fzcap(s->params, zmax, any_unknown);
    return check_match_counters(s, v.fzcap);
}

static int check_batch_args(const SageSpectrumBatch* b) {
    if (b->n_spectra && (!b->peak_off || !b->precursor_mz || !b->precursor_charge || !b->total_ion_current))
        return fail(SAGE_HIP_ERR_INVALID, "missing required spectrum arrays");
    const uint64_t total = b->n_spectra ? b->peak_off[b->n_spectra] - b->peak_off[0] : 0;
    if (total && (!b->masses || !b->intensities)) return fail(SAGE_HIP_ERR_INVALID, "missing peak arrays");
    return SAGE_HIP_OK;
}

// A resident batch knows EXACTLY whether any of its precursor windows exceeds the narrow kernel's slot capacity: one small
// kernel over its spectra behind the upload (window_max_kernel: the partition points of IndexedDatabase::query per query),
// four bytes back with the upload's synchronisation.  The sample-based estimate (choose_probe) said "maybe" for any batch that
// is dense where peptides are dense — e.g. the shard of a rank that owns the light end of the mass axis — and a "maybe" costs the
// step ten empty launches, the second part's overlap, and a longer host turn-around (round 5: 0.75 against 0.69 ms per 62 500
// C3 spectra).  The streaming pipeline keeps the estimate (it decides before the spectra are on the device).
static int exact_window_check(SageScorer* s, SageDeviceBatch* d, hipStream_t st) {
    if (getenv("SAGE_HIP_ASSUME_NARROW") || d->n == 0) return SAGE_HIP_OK;  // (tests force the estimate's answer)
    HIP_TRY(s->win_max.reserve(1));
    HIP_TRY(hipMemsetAsync(s->win_max.p, 0, 4, st));
    launch_window_max(s->dev, d->view, s->db->view.pep_mono, s->db->view.np, s->win_max.p, st);
    HIP_TRY(hipGetLastError());
    uint32_t widest = 0;
    HIP_TRY(hipMemcpyAsync(&widest, s->win_max.p, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    d->maybe_wide = widest > s->dev.wcap;
    d->widest = widest;
    return SAGE_HIP_OK;
}

int sage_hip_batch_upload(SageScorer* s, const SageSpectrumBatch* b, SageDeviceBatch** out) {
    if (!s || !b || !out) return fail(SAGE_HIP_ERR_INVALID, "null argument");
    int rc = check_batch_args(b);
    if (rc != SAGE_HIP_OK) return rc;
    std::lock_guard<std::mutex> lock(s->mu);
    HIP_TRY(hipSetDevice(s->db->device));
    auto d = std::make_unique<SageDeviceBatch>();
    d->device = s->db->device;
    HIP_TRY(d->up_done.create(false));
    const uint32_t n = b->n_spectra;
    const WindowEstimate est = choose_probe(s, n, b->precursor_mz, b->precursor_charge, b->isolation_lo, b->isolation_hi);
    rc = stage_and_upload(s, d.get(), b, 0, n, is_page_locked(b->masses) && is_page_locked(b->intensities), est, s->up_stream);
    if (rc != SAGE_HIP_OK) return rc;
    HIP_TRY(hipStreamSynchronize(s->up_stream));
    rc = exact_window_check(s, d.get(), s->up_stream);
    if (rc != SAGE_HIP_OK) return rc;
    *out = d.release();
    return SAGE_HIP_OK;
}

int sage_hip_batch_process_upload(SageScorer* s, const SageRawBatch* raw, uint64_t take_top_n, int deisotope,
                                  float min_deisotope_mz, uint32_t min_peaks, SageDeviceBatch** out, uint32_t* out_npeaks) {
    if (!s || !raw || !out) return fail(SAGE_HIP_ERR_INVALID, "null argument");
    const uint32_t n = raw->n_spectra;
    if (n && (!raw->peak_off || !raw->precursor_mz || !raw->precursor_charge))
        return fail(SAGE_HIP_ERR_INVALID, "missing required spectrum arrays");
    if (take_top_n == 0 || take_top_n > 0xFFFFu) return fail(SAGE_HIP_ERR_INVALID, "take_top_n must be in [1, 65535]");
    std::lock_guard<std::mutex> lock(s->mu);
    HIP_TRY(hipSetDevice(s->db->device));
    auto d = std::make_unique<SageDeviceBatch>();
    d->device = s->db->device;
    HIP_TRY(d->up_done.create(false));
    const uint64_t total = n ? raw->peak_off[n] : 0;
    uint32_t rcap = 1;
    for (uint32_t i = 0; i < n; i++) {
        if (raw->peak_off[i + 1] < raw->peak_off[i]) return fail(SAGE_HIP_ERR_INVALID, "peak_off is not monotone");
        rcap = std::max<uint32_t>(rcap, (uint32_t)(raw->peak_off[i + 1] - raw->peak_off[i]));
    }
    if (total && (!raw->mz || !raw->intensities)) return fail(SAGE_HIP_ERR_INVALID, "missing peak arrays");
    // the LDS instance of the kernel takes spectra up to PROCESS_LDS_PEAKS raw peaks (what three workgroups per CU can hold);
    // larger ones go through the global-workspace instance, whatever their size
    constexpr uint32_t PROCESS_LDS_PEAKS = 2048;
    const uint32_t big_cap = rcap;
    std::vector<uint32_t> big;
    if (rcap > PROCESS_LDS_PEAKS) {
        for (uint32_t i = 0; i < n; i++)
            if (raw->peak_off[i + 1] - raw->peak_off[i] > PROCESS_LDS_PEAKS) big.push_back(i);
        rcap = PROCESS_LDS_PEAKS;
    }
    uint32_t rpow2 = 1;
    while (rpow2 < rcap) rpow2 <<= 1;
    HIP_TRY((hipError_t)process_kernel_prepare(160 * 1024));
    const uint32_t stride = (uint32_t)std::min<uint64_t>(take_top_n, big_cap);
    DevBuf<uint64_t> raw_off;
    DevBuf<float> raw_mz, raw_int, sm, si;
    DevBuf<uint8_t> zbuf;
    DevBuf<uint32_t> cnt;
    HIP_TRY(raw_off.upload(raw->peak_off, n ? (size_t)n + 1 : 0));
    HIP_TRY(raw_mz.upload(raw->mz, total));
    HIP_TRY(raw_int.upload(raw->intensities, total));
    HIP_TRY(zbuf.upload(raw->precursor_charge, n));
    HIP_TRY(sm.alloc((size_t)n * stride));
    HIP_TRY(si.alloc((size_t)n * stride));
    HIP_TRY(cnt.alloc(n));
    HIP_TRY(d->tic.alloc(n));
    launch_process(n, raw_off.p, raw_mz.p, raw_int.p, zbuf.p, (uint32_t)take_top_n, deisotope != 0, min_deisotope_mz, rcap, rpow2,
                   stride, sm.p, si.p, d->tic.p, cnt.p, s->stream);
    HIP_TRY(hipGetLastError());
    DevBuf<uint32_t> big_list;
    DevBuf<unsigned char> big_ws;
    if (!big.empty()) {
        // Groups of similar size, each with slices sized for ITS largest spectrum, launched one after the other over one
        // workspace of bounded size (stream order makes the reuse safe): a batch with one 50 000-peak outlier among thousands of
        // 3 000-peak spectra must not ask for (number of big spectra) x (the outlier's slice).
        std::sort(big.begin(), big.end(), [&](uint32_t a, uint32_t b) {
            const uint64_t na = raw->peak_off[a + 1] - raw->peak_off[a], nb = raw->peak_off[b + 1] - raw->peak_off[b];
            return na != nb ? na < nb : a < b;
        });
        auto slice_of = [&](uint32_t spec, uint32_t* cap_out, uint32_t* pow2_out) {
            const uint32_t cap = (uint32_t)(raw->peak_off[spec + 1] - raw->peak_off[spec]);
            uint32_t p2 = 1;
            while (p2 < cap) p2 <<= 1;
            if (cap_out) *cap_out = cap;
            if (pow2_out) *pow2_out = p2;
            return process_lds_bytes(cap, p2);
        };
        const size_t budget = (size_t)1 << 30;  // bytes of workspace per launch (a single larger spectrum still gets its slice)
        struct Group { size_t first, count; uint32_t cap, pow2; size_t slice; };
        std::vector<Group> groups;
        for (size_t i = 0; i < big.size();) {
            // grow the group while (members) x (slice of the candidate member, the largest so far) fits the budget
            size_t j = i;
            uint32_t cap = 0, p2 = 0;
            size_t slice = 0;
            while (j < big.size()) {
                uint32_t c, q;
                const size_t sl = slice_of(big[j], &c, &q);
                if (j > i && (j - i + 1) * sl > budget) break;
                cap = c; p2 = q; slice = sl;
                j++;
            }
            groups.push_back(Group{i, j - i, cap, p2, slice});
            i = j;
        }
        size_t ws_bytes = 0;
        for (const Group& g : groups) ws_bytes = std::max(ws_bytes, g.count * g.slice);
        HIP_TRY(big_list.upload(big.data(), big.size()));
        if (hipError_t e = big_ws.alloc(ws_bytes); e != hipSuccess) {
            (void)hipGetLastError();
            return fail(e == hipErrorOutOfMemory ? SAGE_HIP_ERR_OOM : SAGE_HIP_ERR_HIP,
                        "preprocessing workspace of " + std::to_string(ws_bytes >> 20) + " MiB for a spectrum of " + std::to_string(big_cap) +
                            " raw peaks: " + hipGetErrorString(e) + " (sage_hip_process_ms2 preprocesses a spectrum on the host)");
        }
        for (const Group& g : groups) {
            launch_process_big((uint32_t)g.count, big_list.p + g.first, big_ws.p, raw_off.p, raw_mz.p, raw_int.p, zbuf.p, (uint32_t)take_top_n,
                               deisotope != 0, min_deisotope_mz, g.cap, g.pow2, stride, sm.p, si.p, d->tic.p, cnt.p, s->stream);
            HIP_TRY(hipGetLastError());
        }
    }
    std::vector<uint32_t> counts(n);
    HIP_TRY(hipMemcpyAsync(counts.data(), cnt.p, (size_t)n * 4, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (out_npeaks) std::copy(counts.begin(), counts.end(), out_npeaks);
    std::vector<uint64_t> off((size_t)n + 1, 0);
    uint32_t pcap = 1, zmax = 0;
    bool any_unknown = false;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t c = counts[i] >= min_peaks ? counts[i] : 0;  // runner.rs:313: too few peaks -> not searched
        off[i + 1] = off[i] + c;
        pcap = std::max(pcap, c);
        zmax = std::max<uint32_t>(zmax, raw->precursor_charge[i]);
        any_unknown = any_unknown || raw->precursor_charge[i] == 0;
    }
    HIP_TRY(d->peak_off.upload(off.data(), n ? (size_t)n + 1 : 0));
    HIP_TRY(d->masses.alloc(off[n]));
    HIP_TRY(d->intensities.alloc(off[n]));
    launch_compact(n, d->peak_off.p, stride, sm.p, si.p, d->masses.p, d->intensities.p, s->stream);
    HIP_TRY(hipGetLastError());
    // precursor-side arrays and the launch schedule
    d->n = n;
    HIP_TRY(d->precursor_mz.upload(raw->precursor_mz, n));
    HIP_TRY(d->charge.upload(raw->precursor_charge, n));
    const bool has_iso = raw->isolation_lo && raw->isolation_hi;
    if (has_iso) {
        HIP_TRY(d->iso_lo.upload(raw->isolation_lo, n));
        HIP_TRY(d->iso_hi.upload(raw->isolation_hi, n));
    }
    if (raw->scan_start_time) HIP_TRY(d->rt.upload(raw->scan_start_time, n));
    if (raw->inverse_ion_mobility) HIP_TRY(d->ims.upload(raw->inverse_ion_mobility, n));
    if (raw->file_id) HIP_TRY(d->file_id.upload(raw->file_id, n));
    HIP_TRY(d->order.alloc(n));
    HIP_TRY(d->sort_a.alloc(n));
    HIP_TRY(d->sort_b.alloc(n));
    HIP_TRY(d->sort_idx.alloc(n));
    const size_t sort_bytes = schedule_temp_bytes(n);
    HIP_TRY(d->sort_tmp.alloc(sort_bytes));
    HIP_TRY((hipError_t)schedule_on_device(n, d->precursor_mz.p, d->charge.p, s->params.min_precursor_charge, d->sort_a.p, d->sort_b.p,
                                           d->sort_idx.p, d->order.p, d->sort_tmp.p, sort_bytes, s->stream));
    const bool use_sched = !getenv("SAGE_HIP_NO_SCHED");
    if (use_sched) {
        HIP_TRY(d->sched.alloc(2 * (size_t)n));
        schedule_records_on_device(n, d->order.p, d->peak_off.p, d->precursor_mz.p, d->charge.p, has_iso ? d->iso_lo.p : nullptr,
                                   has_iso ? d->iso_hi.p : nullptr, d->sched.p, s->stream);
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    DevBatchView& v = d->view;
    v = DevBatchView{};
    v.n = n;
    v.peak_off = d->peak_off.p;
    v.masses = d->masses.p;
    v.intensities = d->intensities.p;
    v.precursor_mz = d->precursor_mz.p;
    v.precursor_charge = d->charge.p;
    v.isolation_lo = has_iso ? d->iso_lo.p : nullptr;
    v.isolation_hi = has_iso ? d->iso_hi.p : nullptr;
    v.tic = d->tic.p;
    v.rt = raw->scan_start_time ? d->rt.p : nullptr;
    v.ims = raw->inverse_ion_mobility ? d->ims.p : nullptr;
    v.file_id = raw->file_id ? d->file_id.p : nullptr;
    v.order = d->order.p;
    v.sched = use_sched && n ? d->sched.p : nullptr;
    const WindowEstimate est = choose_probe(s, n, raw->precursor_mz, raw->precursor_charge, raw->isolation_lo, raw->isolation_hi);
    v.probe = est.probe;
    d->maybe_wide = est.maybe_wide;
    v.pcap = pcap;
    v.fzcap = batch_fzcap(s->params, zmax, any_unknown);
    if (int rc = check_match_counters(s, v.fzcap); rc != SAGE_HIP_OK) return rc;
    if (int rc = exact_window_check(s, d.get(), s->stream); rc != SAGE_HIP_OK) return rc;
    *out = d.release();
    return SAGE_HIP_OK;
}

int sage_hip_batch_download(SageDeviceBatch* b, uint64_t* peak_off, float* masses, float* intensities, float* tic) {
    if (!b || !peak_off) return fail(SAGE_HIP_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(b->device));
    if (b->n) HIP_TRY(hipMemcpy(peak_off, b->view.peak_off, ((size_t)b->n + 1) * 8, hipMemcpyDeviceToHost));
    else peak_off[0] = 0;
    const uint64_t total = b->n ? peak_off[b->n] : 0;
    if (masses && total) HIP_TRY(hipMemcpy(masses, b->view.masses, total * 4, hipMemcpyDeviceToHost));
    if (intensities && total) HIP_TRY(hipMemcpy(intensities, b->view.intensities, total * 4, hipMemcpyDeviceToHost));
    if (tic && b->n) HIP_TRY(hipMemcpy(tic, b->view.tic, (size_t)b->n * 4, hipMemcpyDeviceToHost));
    return SAGE_HIP_OK;
}

void sage_hip_batch_free(SageDeviceBatch* b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    delete b;
}

// ---- scoring ---------------------------------------------------------------------------------------------------------------
static uint64_t arena_entries_for(const SageScorer* s, uint32_t n) {
    // 16 Ki entries = 64 KiB per spectrum on average + two 64 Ki-entry chunks per resident workgroup (each takes its arena space
    // a chunk at a time); an exhausted arena is detected, and the batch is then scored in smaller pieces
    uint64_t e = std::max<uint64_t>((uint64_t)n * 16384, 16ull << 20) + (uint64_t)std::min(n, std::max(s->tile_blocks, s->tile_blocks8)) * (2u << 16);
    e = std::min<uint64_t>(e, 0xFFFFFFF0ull);
    if (const char* v = getenv("SAGE_HIP_ARENA_MB")) e = std::min<uint64_t>((uint64_t)std::max(1, atoi(v)) << 18, 0xFFFFFFF0ull);
    return e;
}

// The device working set of a launch of n spectra.  `lane`: which of the scorer's two sets (the streaming pipeline's second compute
// lane never takes the large-window path, so it never owns that path's buffers — the candidate arena alone is 64 KiB per
// spectrum).  `wide`: the large-window kernels will be launched (lane 0 only).  `st`: the stream the kernels will run on.
static int ensure_work(SageScorer* s, uint32_t n, int lane, bool wide, hipStream_t st) {
    WorkSet& w = lane ? s->ws2 : s->ws;
    if (n > w.cap_n) {
        HIP_TRY(w.cand.reserve((size_t)n * s->dev.kmax + 64));  // (+ 64: rescore_kernel reads a full wavefront's worth of every row)
        HIP_TRY(w.cand_len.reserve(n));
        HIP_TRY(w.totals.reserve((size_t)n * 2));
        HIP_TRY(w.status.reserve(n));
        HIP_TRY(w.queue.reserve(n));
        HIP_TRY(w.retry.reserve(n));
        HIP_TRY(w.item_of.reserve(n));
        HIP_TRY(w.ready.reserve(n));
        // (no launch has the epoch 0.  On the stream of the kernels that read it: the scorer's streams do not synchronise with
        // the null stream, and a plain hipMemset could land after a producer's epoch store)
        HIP_TRY(hipMemsetAsync(w.ready.p, 0, (size_t)w.ready.n * 4, st));
        HIP_TRY(hipStreamSynchronize(st));  // (growth is rare; the parts of a resident step start on their own streams without waiting for this one)
        w.epoch = 0;
        w.cap_n = n;
    }
    if (s->fast_ties && n > w.cap_tie) {
        // (n rows of ~2 KB: 1 GB for a 500 000-spectrum resident batch.  The search does not need them — without the rows every
        // tie takes the exact retry pass — so running out of memory HERE switches fast ties off instead of failing the call.)
        const hipError_t e = w.cnt_store.reserve((size_t)n * s->cnt_stride);
        if (e == hipErrorOutOfMemory) {
            (void)hipGetLastError();
            s->fast_ties = false;
        } else {
            HIP_TRY(e);
            w.cap_tie = n;
        }
    }
    if (wide && lane == 0 && n > w.cap_wide) {
        // large-window pipeline: per-query records, verbatim slots, replayed heaps, and the candidate arena
        HIP_TRY(w.qrec.reserve((size_t)n * s->qmax));
        HIP_TRY(w.seeds.reserve((size_t)n * s->qmax * s->kstride));
        HIP_TRY(w.qres.reserve((size_t)n * s->qmax * s->kstride));
        const uint64_t arena_entries = arena_entries_for(s, n);
        if (arena_entries > w.arena.n) HIP_TRY(w.arena.alloc(arena_entries));
        w.cap_wide = n;
    }
    return SAGE_HIP_OK;
}
// PSM counts always; a device-side record buffer only when the records do not go straight to the caller's page-locked array
static int ensure_out(SageScorer* s, OutSet& o, uint32_t n, bool need_features) {
    HIP_TRY(o.out_count.reserve(n));
    if (need_features) HIP_TRY(o.features.reserve((size_t)n * s->params.report_psms));
    return SAGE_HIP_OK;
}

static DevWork make_work(SageScorer* s, OutSet& o, int pass, int lane = 0) {
    DevWork w{};
    WorkSet& ws = lane ? s->ws2 : s->ws;
    w.cand = ws.cand.p;
    w.cand_len = ws.cand_len.p;
    w.totals = ws.totals.p;
    w.status = ws.status.p;
    w.n_deferred = o.counters.p + (size_t)pass * CTR_COUNT;
    w.queue = ws.queue.p;
    w.retry = ws.retry.p;
    w.item_of = ws.item_of.p;
    w.ready = ws.ready.p;
    w.epoch = ws.epoch;
    w.search_lag = s->search_lag;
    w.replay_split = s->replay_split;
    w.kstride = s->kstride;
    w.reuse = 0;
    w.arena_ptr = w.n_deferred + CTR_ARENA_PTR;
    w.cnt_store = nullptr;  // (enqueue_compute switches the cheap ties on for the first pass of a production search)
    w.cnt_stride = s->cnt_stride;
    w.tile_blocks = s->tile_blocks;
    w.qrec = ws.qrec.p;
    w.seeds = ws.seeds.p;
    w.qres = ws.qres.p;
    w.arena = ws.arena.p;
    w.arena_cap = (uint32_t)ws.arena.n;
    w.qmax = s->qmax;
    w.tile_shift = s->db->view.tile_shift;
    w.dbg = s->dbg.p;
    return w;
}

enum { MODE_SCORE = 0,  // order-free trims, then the exact retry pass over the spectra whose reported ranks tie
       MODE_EXACT = 1,  // every trim replays bounded_min_heapify: one pass
       MODE_FAST = 2 }; // order-free trims only (quick_score: which peptides survive does not depend on heap layouts)

// Enqueue the kernels of one batch on `st` — nothing here waits for the device.
//   MODE_SCORE with rescoring, the production path.  First pass, order-free trims: prelim_kernel scores the narrow windows and
//   queues the others; when `wide`, the large-window kernels take the queue; rescore_kernel reports every spectrum except those
//   whose reported ranks tie in hyperscore, which it lists.  Exact retry pass over that list, launched unconditionally — the
//   list and its COUNT live on the device (the first pass's CTR_RETRY counter), so no host round trip separates the passes, and
//   with no tied spectrum its blocks exit at once: narrow_kernel (matching with bounded_min_heapify replayed + rescoring, one
//   launch) reports the narrow ones and queues the others; when `wide`, the large-window kernels and rescore_kernel follow.
//   `wide` == false skips the ten launches of the large-window path: the caller checks the queue counter afterwards and
//   repeats the batch with wide == true if the guess was wrong.  SAGE_HIP_FUSED=1: narrow_kernel as the first pass, too.
//   Other modes: preliminary kernel, large-window kernels, rescoring kernel.
// `rec`: where the PSM records go — device memory (null: the OutSet's buffer) or the device-side view of page-locked host memory.
// `list_off` / `count_buf`: a part of a batch scored next to other parts (score_resident_locked): its queue / retry lists start
// at that offset of the shared arrays, and the PSM counts of all parts go to one buffer (they are indexed by spectrum).
// `lane`: which of the scorer's two working sets (a lane-1 batch never takes the large-window path: the replay kernels' side
// stream is lane 0's).
static int enqueue_compute(SageScorer* s, const DevBatchView& view_in, OutSet& o, bool with_rescore, int mode, hipStream_t st,
                           SageFeature* rec = nullptr, bool wide = true, uint32_t list_off = 0, uint32_t* count_buf = nullptr,
                           int lane = 0, int phase = 0, uint32_t widest = 0xFFFFFFFFu) {
    // `phase` (resident steps, score_resident_locked): 0 both passes; 1 the first pass only — the exact retry pass is launched
    // later if it turns out to have anything to do; 2 that retry pass alone (the first pass of this very launch has completed).
    if (lane && wide) return fail(SAGE_HIP_ERR_INTERNAL, "large windows on the second working set");
    WorkSet& wset = lane ? s->ws2 : s->ws;
    DevScorer sc = s->dev;
    if (wide && !s->sched_desc_forced) sc.xcd_chunk &= 0x7FFFFFFFu;  // (heaviest-first is the narrow search's default: scorer_init)
    DevBatchView view = view_in;
    // (the stream variant of the preliminary kernels keeps a window per (peak, fragment charge) in LDS, the probe variant the peak
    // masses only: a batch of very large spectra is probed whatever the estimate said)
    if (!view.probe && (size_t)view.fzcap * view.pcap * 8 > 48 * 1024) view.probe = 1;
    if (!view.probe) {  // (the stream variant reads the peptide-major fragment list, which an index keeps only once asked for)
        const int rc_pm = ensure_pm_frag(s->db);
        if (rc_pm != SAGE_HIP_OK) return rc_pm;
    }
    const bool production = mode == MODE_SCORE && with_rescore;
    const bool fused = s->fused && production;
    if (!production) wide = true;
    int rc = ensure_work(s, view.n, lane, wide, st);
    if (rc != SAGE_HIP_OK) return rc;
    rc = ensure_out(s, o, view.n, rec == nullptr);
    if (rc != SAGE_HIP_OK) return rc;
    if (!rec) rec = o.features.p;
    if (!count_buf) count_buf = o.out_count.p;
    const bool one_launch = production && !fused && s->one_launch;
    // Lists wider than a wavefront (report_psms > 32) live in LDS while they fit a compute unit's 160 KB; a configuration whose
    // lists, heaps or per-candidate arrays do not — report_psms in the hundreds x many precursor-window queries per spectrum,
    // report_psms beyond ~500, thousands of peaks on top — gets them in a global-memory workspace, a slice per workgroup of the
    // (then capped) grids.  Slower again, and never a refusal.
    bool huge = false;
    if (sc.big_path) {
        const size_t cap = (size_t)160 * 1024;
        huge = prelim_lds_bytes(sc, view) > cap || rescore_lds_bytes(sc, view, s->db->max_ions, true) > cap || assemble_lds_bytes(sc) > cap ||
               (size_t)s->kstride * 8 + 256 > cap || getenv("SAGE_HIP_FORCE_HUGE") != nullptr;  // (tests force the workspace on small lists)
    }
    const size_t lds_p = sc.big_path ? prelim_lds_bytes(sc, view, huge)
                                     : std::max(production ? std::max(narrow_lds_bytes(sc, view), one_launch ? search_lds_bytes(sc, view) : (size_t)0) : (size_t)0, prelim_lds_bytes(sc, view)),
                 lds_r = rescore_lds_bytes(sc, view, s->db->max_ions, true, huge);
    // the large-window count kernel keeps a window per (peak, fragment charge) in LDS next to a tile's counters; when that does not
    // fit a compute unit, the instance with the windows in global memory takes the batch (tile_count_wing_kernel)
    const bool wing = wide && tile_lds_bytes(s->db->view, sc, view) > 160 * 1024;
    const size_t lds_t = wide ? tile_lds_bytes(s->db->view, sc, view, false, wing) : 0;
    // (every per-spectrum kernel may take a whole CU's LDS: spectrum_kernel_prepare / bigk_kernel_prepare at scorer creation)
    const size_t lds_cap = (size_t)160 * 1024;
    const size_t lds_a = sc.big_path && !huge ? assemble_lds_bytes(sc) : 0;
    if (lds_p > lds_cap || lds_r > lds_cap || lds_a > lds_cap || lds_t > 160 * 1024)
        return fail(SAGE_HIP_ERR_UNSUPPORTED,
                    sc.big_path ? "candidate lists too long for the LDS of a compute unit (report_psms x precursor-window queries per spectrum, or peaks x "
                                   "fragment charges): lower report_psms or narrow the charge / isotope-error ranges"
                                 : "spectrum too large for the LDS of a compute unit (~15 000 peaks per processed spectrum): lower max_peaks");
    if (huge) {
        // (a slice per workgroup of the capped grids — and a batch of fewer spectra launches fewer workgroups: ADVICE r05)
        // (the per-query kernels launch min(spectra x queries per spectrum, cap) workgroups, the per-spectrum ones fewer)
        const uint64_t blocks = std::max<uint64_t>((uint64_t)view.n * std::max<uint32_t>(s->qmax, 1u), 1u);
        const size_t need = (size_t)std::min<uint64_t>(huge_grid(), blocks) * huge_stride_bytes(sc);
        if (wset.hugebuf.n < need) {
            HIP_TRY(hipStreamSynchronize(st));
            HIP_TRY(wset.hugebuf.reserve(need));
        }
    }
    if (wing) {  // [tile_blocks][2][fzcap * pcap] floats
        const size_t need = (size_t)std::max(s->tile_blocks, s->tile_blocks8) * 2 * view.fzcap * view.pcap;
        if (wset.winbuf.n < need) {
            HIP_TRY(hipStreamSynchronize(st));
            HIP_TRY(wset.winbuf.reserve(need));
        }
    }
    o.two_pass = production && !(fused && !wide);  // (the fused first pass settles the ties of narrow spectra itself)
    o.with_rescore = with_rescore;
    o.fused = fused || one_launch;
    o.wide_launched = wide;
    o.n = view.n;
    DevBatchView v2 = view;
    v2.sched = nullptr;  // (the retry list is not the schedule)
    v2.order = wset.retry.p + list_off;  // filled by the rescoring kernel of the first pass, in no particular order
    v2.n_dev = o.counters.p + CTR_RETRY;
    if (one_launch && ++wset.epoch == 0) wset.epoch = 1;
    DevWork w1 = make_work(s, o, 0, lane), w2 = make_work(s, o, 1, lane);
    for (DevWork* w : {&w1, &w2}) {
        w->queue += list_off;
        w->retry += list_off;
    }
    const bool fast_ties = s->fast_ties && production && !fused && !one_launch && o.two_pass;
    if (fast_ties) {  // (each part of a step owns the rows / entries [list_off, list_off + n) of these arrays)
        w1.cnt_store = wset.cnt_store.p + (size_t)list_off * s->cnt_stride;  // (rows by schedule position within the part)
    }
    if (wing) w1.winbuf = w2.winbuf = wset.winbuf.p;
    o.huge = huge;
    if (huge) {
        w1.hugebuf = w2.hugebuf = wset.hugebuf.p;
        w1.huge_stride = w2.huge_stride = (uint32_t)huge_stride_bytes(sc);
    }
    if (production && s->cnt8 && !wing) {  // (only a pass that is followed by the retry pass may count in u8)
        w1.cnt8 = 1;
        w1.tile_blocks = s->tile_blocks8;
    }
    if (production && s->reuse_counts) {  // the retry pass replays from the first pass's counts, appending to the same arena
        w2.reuse = 1;
        w2.arena_ptr = w1.arena_ptr;
    }
    const SideStream side{s->side_stream, s->side_fork.e, s->side_join.e};
    DevScorer sc1 = sc, sc2 = sc;
    sc1.exact = mode == MODE_EXACT ? 1u : 0u;
    sc1.fast_log = o.two_pass ? 1u : 0u;  // (an undecided logarithm is settled by the retry pass: launch_rescore)
    sc2.exact = 1u;
    sc2.fast_log = 0u;
    const bool no_timing = !s->timed;  // (sage_hip_scorer_set_timing_interval)
    o.timed = s->timed;
    o.retry_deferred = phase == 1 && o.two_pass;
    if (phase == 2) {
        // the retry list's length: the first pass left it in the counter block, the epilogue sent it home and zeroed the block
        HIP_TRY(hipMemcpyAsync(o.counters.p + CTR_RETRY, o.h_counters + CTR_RETRY, 4, hipMemcpyHostToDevice, st));
        o.counters_clean = false;
        if (!no_timing) HIP_TRY(hipEventRecord(o.ev[2].e, st));
        goto retry_pass;
    }
    // (score_resident's epilogue kernel leaves the counters zeroed for the next step: one command less ahead of the first kernel)
    if (!o.counters_clean) HIP_TRY(hipMemsetAsync(o.counters.p, 0, 2 * CTR_COUNT * 4, st));
    o.counters_clean = false;
    if (!no_timing) HIP_TRY(hipEventRecord(o.ev[0].e, st));
    if (fused)
        launch_narrow(s->db->view, sc1, view, w1, s->lnfact.p, (uint32_t)s->lnfact.n, rec, count_buf, st);
    else if (one_launch)
        launch_search(s->db->view, sc1, view, w1, s->lnfact.p, (uint32_t)s->lnfact.n, rec, count_buf, st);
    else {
        // With the large-window kernels behind it the preliminary kernel may only MARK the spectra it hands over, one small kernel
        // building their queue in ascending precursor mass (DevWork::queue_later) — where the batch's windows are known to lie
        // within a few tiles (a wide-window / DIA search).  One returning atomic per spectrum on the queue's counter costs
        // ~11 ns each (2.3 ms of C5's step, where every spectrum is handed over), and neighbours in the sorted queue share their
        // tiles: C5 44.3 -> 43.2 ms.  Not for windows of dozens of tiles (an open search): there the count kernel runs 6 % SLOWER
        // behind ANY sorted or regularly permuted queue than behind the atomics' arrival order (C4 33.6 -> 35.8 ms of
        // tile_count8_kernel, the same binary: not understood) — scripts/experiments/r06_lab/RESULTS.md, r8e - r8m.
        // (the line between the two: 16 tiles — C5's widest window, charge 4 of a 12 Da isolation window, holds ~4 tiles' worth of
        // candidates, C4's +-500 Da ~55)
        bool later = wide && widest <= (16u << s->db->view.tile_shift);
        if (const char* e = getenv("SAGE_HIP_QUEUE_LATER")) later = wide && atoi(e) != 0;  // (tests and experiments force either way)
        w1.queue_later = later ? 1u : 0u;
        launch_prelim(s->db->view, sc1, view, w1, st);
        if (later) launch_queue(sc1, view, w1, st);
    }
    HIP_TRY(hipGetLastError());  // (a failed launch must not let the kernels downstream of it run on stale records)
    if (wide) {
        launch_prelim_tile(s->db->view, sc1, view, w1, st, &side);
        HIP_TRY(hipGetLastError());
    }
    if (!no_timing) HIP_TRY(hipEventRecord(o.ev[1].e, st));
    if (with_rescore && (wide || !(fused || one_launch)))  // (behind search_kernel / the fused kernel: only the spectra the large-window kernels assembled)
        launch_rescore(s->db->view, sc1, view, w1, s->lnfact.p, (uint32_t)s->lnfact.n, s->db->max_ions, rec, count_buf, nullptr, st);
    if (!no_timing) HIP_TRY(hipEventRecord(o.ev[2].e, st));
retry_pass:
    if (o.two_pass && phase != 1) {
        launch_narrow(s->db->view, sc2, v2, w2, s->lnfact.p, (uint32_t)s->lnfact.n, rec, count_buf, st);
        HIP_TRY(hipGetLastError());
        if (wide) {
            launch_prelim_tile(s->db->view, sc2, v2, w2, st, &side);
            HIP_TRY(hipGetLastError());
        }
        if (wide) {  // (without the large-window kernels the retry pass is ONE launch: no marker inside it)
            if (!no_timing) HIP_TRY(hipEventRecord(o.ev[3].e, st));
            launch_rescore(s->db->view, sc2, v2, w2, s->lnfact.p, (uint32_t)s->lnfact.n, s->db->max_ions, rec, count_buf, nullptr, st);
        }
        if (!no_timing) HIP_TRY(hipEventRecord(o.ev[4].e, st));
    }
    HIP_TRY(hipGetLastError());
    o.in_flight = true;
    return SAGE_HIP_OK;
}

// counters + timing of a finished batch (its comp_done / down_done event has completed); accumulates into s->timing.
// *redo_wide: the batch was launched without the large-window kernels and turned out to need them — nothing is accumulated.
static int collect(SageScorer* s, OutSet& o, bool* arena_overflow, bool* redo_wide = nullptr) {
    o.in_flight = false;
    const uint32_t* c1 = o.h_counters;
    const uint32_t* c2 = o.h_counters + CTR_COUNT;
    if (!o.wide_launched && c1[CTR_QUEUED]) {
        if (redo_wide) {
            *redo_wide = true;
            return SAGE_HIP_OK;
        }
        return fail(SAGE_HIP_ERR_INTERNAL, "spectra queued for the large-window kernels, which were not launched");
    }
    float a = 0, r = 0, a2 = 0, r2 = 0;
    if (!o.timed) goto counters;
    HIP_TRY(hipEventElapsedTime(&a, o.ev[0].e, o.ev[1].e));
    HIP_TRY(hipEventElapsedTime(&r, o.ev[1].e, o.ev[2].e));
    if (o.two_pass && o.retry_deferred) {
        // (the retry pass has not run: collect_retry adds its share if it does)
    } else if (o.two_pass && o.wide_launched) {
        HIP_TRY(hipEventElapsedTime(&a2, o.ev[2].e, o.ev[3].e));
        HIP_TRY(hipEventElapsedTime(&r2, o.ev[3].e, o.ev[4].e));
    } else if (o.two_pass) {
        HIP_TRY(hipEventElapsedTime(&a2, o.ev[2].e, o.ev[4].e));
    }
counters:
    SageTiming& t = s->timing;
    t.prelim_ms += a + a2;
    t.rescore_ms += o.with_rescore ? r + r2 : 0.f;
    t.total_ms += a + a2 + (o.with_rescore ? r + r2 : 0.f);
    t.retry_ms += a2 + r2;
    t.n_launches += 1 + (o.wide_launched ? 4 : 0) + (o.with_rescore && (o.wide_launched || !o.fused) ? 1 : 0) +
                    (o.two_pass && !o.retry_deferred ? 1 + (o.wide_launched ? 5 : 0) : 0);
    t.n_wide += c1[CTR_QUEUED];
    t.arena_entries = std::max(t.arena_entries, std::max(c1[CTR_ARENA_PTR], c2[CTR_ARENA_PTR]));
    t.n_retry += o.two_pass ? c1[CTR_RETRY] : 0;
    t.n_tied += c1[CTR_TIED];
    for (uint32_t k = 0; k < CTR_TIE_STRIPES; k++) t.n_tied += c1[CTR_TIE_STRIPE0 + k * CTR_TIE_STRIDE];
    if (c1[CTR_ARENA_OVERFLOW] || c2[CTR_ARENA_OVERFLOW]) {
        if (arena_overflow) {
            *arena_overflow = true;
            return SAGE_HIP_OK;
        }
        return fail(SAGE_HIP_ERR_UNSUPPORTED, "large-window candidate arena exhausted (" + std::to_string(s->ws.arena.n >> 18) +
                                                  " MiB): score this batch in smaller pieces or raise SAGE_HIP_ARENA_MB");
    }
    if (c1[CTR_LIST_OVERFLOW] || c2[CTR_LIST_OVERFLOW])
        return fail(SAGE_HIP_ERR_UNSUPPORTED, "preliminary candidate list capacity exceeded (" +
                                                  std::to_string(c1[CTR_LIST_OVERFLOW] + c2[CTR_LIST_OVERFLOW]) + " spectra)");
    return SAGE_HIP_OK;
}

// ... of a retry pass that was launched by itself, behind a first pass that has been collected (enqueue_compute: phase 2)
static int collect_retry(SageScorer* s, OutSet& o) {
    o.in_flight = false;
    o.retry_deferred = false;
    const uint32_t* c2 = o.h_counters + CTR_COUNT;
    SageTiming& t = s->timing;
    if (o.timed) {
        float a2 = 0, r2 = 0;
        if (o.wide_launched) {
            HIP_TRY(hipEventElapsedTime(&a2, o.ev[2].e, o.ev[3].e));
            HIP_TRY(hipEventElapsedTime(&r2, o.ev[3].e, o.ev[4].e));
        } else {
            HIP_TRY(hipEventElapsedTime(&a2, o.ev[2].e, o.ev[4].e));
        }
        t.prelim_ms += a2;
        t.rescore_ms += o.with_rescore ? r2 : 0.f;
        t.total_ms += a2 + (o.with_rescore ? r2 : 0.f);
        t.retry_ms += a2 + r2;
    }
    t.n_launches += 1 + (o.wide_launched ? 5 : 0);
    t.arena_entries = std::max(t.arena_entries, c2[CTR_ARENA_PTR]);
    if (c2[CTR_ARENA_OVERFLOW])
        return fail(SAGE_HIP_ERR_UNSUPPORTED, "large-window candidate arena exhausted (" + std::to_string(s->ws.arena.n >> 18) +
                                                  " MiB): score this batch in smaller pieces or raise SAGE_HIP_ARENA_MB");
    if (c2[CTR_LIST_OVERFLOW])
        return fail(SAGE_HIP_ERR_UNSUPPORTED, "preliminary candidate list capacity exceeded (" + std::to_string(c2[CTR_LIST_OVERFLOW]) + " spectra)");
    return SAGE_HIP_OK;
}

static void reset_timing(SageScorer* s) { s->timing = SageTiming{}; }

// Wait for a stream by polling first: a blocked hipStreamSynchronize wakes up tens of microseconds after the last command — a
// twentieth of a small step.  (Bounded: a long wait falls back to the blocking call.)
static hipError_t wait_for_stream(hipStream_t st) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(st);
        if (e != hipErrorNotReady) return e;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) return hipStreamSynchronize(st);
    }
}

// score a resident batch on the compute stream: kernels, record download, ONE host synchronisation.  Page-locked result
// arrays receive the records straight from the kernels (the stores cross PCIe while the other wavefronts compute: no download
// phase); pageable ones through the device buffer and a copy.
static int score_resident_locked(SageScorer* s, SageDeviceBatch* b, SageFeature* out, uint32_t* out_count) {
    if (b->device != s->db->device) return fail(SAGE_HIP_ERR_INVALID, "batch and scorer live on different devices");
    HIP_TRY(hipSetDevice(s->db->device));
    reset_timing(s);
    // SAGE_HIP_STEP_TRACE=1: host-side clock of a resident step on stderr (microseconds since the call began, and since the
    // previous call returned) — where the host's share of a small step goes (scripts/experiments/r05_lab/gpu_r5i.sh)
    static const bool step_trace = getenv("SAGE_HIP_STEP_TRACE") != nullptr;
    static thread_local std::chrono::steady_clock::time_point t_prev_exit{};
    const auto t_enter = std::chrono::steady_clock::now();
    double t_enq = 0, t_wait = 0;
    auto us_since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); };
    struct TraceOut {
        bool on;
        const std::chrono::steady_clock::time_point* enter;
        std::chrono::steady_clock::time_point* prev;
        const double *enq, *wait;
        ~TraceOut() {
            if (!on) return;
            const auto now = std::chrono::steady_clock::now();
            fprintf(stderr, "[sage_hip] step: %.1f us since the previous call returned; enqueued at %.1f, kernels done at %.1f, returns at %.1f\n",
                    std::chrono::duration<double, std::micro>(*enter - *prev).count(), *enq, *wait,
                    std::chrono::duration<double, std::micro>(now - *enter).count());
            *prev = now;
        }
    } trace_out{step_trace, &t_enter, &t_prev_exit, &t_enq, &t_wait};
    s->timed = s->timing_every != 0 && (s->timing_calls++ % s->timing_every) == 0;
    struct Restore {  // (the other entry points — streaming, quick_score, initial_hits — always record their events)
        SageScorer* s;
        ~Restore() { s->timed = true; }
    } restore{s};
    OutSet& o = s->outs[0];
    SageFeature* direct = s->zero_copy && b->n ? (SageFeature*)device_view_cached(out) : nullptr;
    const size_t rec_bytes = (size_t)b->n * s->params.report_psms * sizeof(SageFeature);
    for (int attempt = 0;; attempt++) {
        // A narrow-search batch in `ways` parts of the launch schedule (consecutive precursor masses), each on its own stream:
        // a part's kernels fill the GPU while another part's kernel starts cold, drains, or waits on the few wavefronts of its
        // retry pass.  Large-window batches go as one (their steps are long; the parts would fight over the candidate arena).
        // Default (no SAGE_HIP_WAYS): two parts for batches up to 196 608 spectra — the shards of a rank of a 3- to 8-GPU
        // strong-scaling run — where a good part of the step is cold starts, tails and the retry pass's handful of wavefronts (C3,
        // 62 500 spectra: -6 %; 125 000 .. 170 000 with the heaviest-first schedule of round 6: -1 .. -1.6 %); one part above, where
        // the kernels' durations are what the roofline is held against.
        const uint32_t want = s->ways ? s->ways : (b->n <= 196608u ? 2u : 1u);
        const uint32_t ways = (!b->maybe_wide && !s->exact_always && b->n >= 8192u * want) ? want : 1u;
        // (sized for the whole batch: the parts of a step index the working set by spectrum and share outs[0]'s count buffer and,
        // if any, its record buffer)
        int rc = ensure_work(s, b->n, 0, b->maybe_wide || s->exact_always, s->stream);
        if (rc != SAGE_HIP_OK) return rc;
        rc = ensure_out(s, o, b->n, direct == nullptr);
        if (rc != SAGE_HIP_OK) return rc;
        SageFeature* const rec = direct ? direct : o.features.p;
        uint32_t* const count_view = s->zero_copy && b->n ? (uint32_t*)device_view_cached(out_count) : nullptr;
        const bool epilogue = count_view != nullptr;  // (page-locked count array: the small results go home in one launch)
        bool reset_done = false;
        // The exact retry pass (spectra with equal hyperscores at a reported rank that the rescoring kernel could not settle itself,
        // undecided logarithms) is one more launch per part behind the rescoring kernel — ~10 us of a step's tail for a list that
        // is empty in most searches (C3, C3T: 0 of 500 000 spectra).  A narrow-search step whose predecessor had nothing to retry
        // does not launch it: the first pass's counters come home with the epilogue, and only if they name retries is the retry
        // pass launched then, behind a second (short) wait.  The first step of a scorer, and any step after one that did retry,
        // launches it unconditionally, as before.
        const bool defer_retry = epilogue && direct && !b->maybe_wide && !s->exact_always && !s->retry_likely && !s->fused && !s->one_launch;
        // No events between the parts' streams: every entry point of a scorer returns with its streams idle, so a part's
        // stream has nothing to wait for at the start (round 4 forked the parts off the first stream with an event: the second
        // part's first kernel started ~30 us after the first's), and at the end every part sends ITS counts and counters home
        // with its own epilogue launch, back to back with its last kernel, and the host waits for each stream in turn (round 4
        // joined the parts on one stream for a single epilogue: ~20 us of cross-stream wake-up behind the last kernel of every
        // step; profiles/r05_small_step_timeline.txt).
        for (uint32_t wy = 0; wy < ways; wy++) {
            const uint32_t start = (uint32_t)((uint64_t)b->n * wy / ways), end = (uint32_t)((uint64_t)b->n * (wy + 1) / ways);
            hipStream_t st = wy ? s->way_stream[wy - 1] : s->stream;
            DevBatchView v = b->view;
            v.order += start;
            if (v.sched) v.sched += 2 * (size_t)start;
            v.n = end - start;
            OutSet& ow = s->outs[wy];
            if (wy == 0 && s->timed) HIP_TRY(hipEventRecord(s->way_begin.e, st));  // (SageTiming::total_ms: first part's start to last part's end)
            // the PSM counts: a step in parts has its kernels store them straight into the caller's page-locked array, as they do
            // the records (4-byte stores spread over the kernels' lifetime cost nothing there; a part's counts are scattered over
            // the array — its spectra are a range of the launch SCHEDULE — and gathering them at the end of the step costs 33 us of
            // scattered stores across the link, joining the parts for one contiguous copy 20 us of cross-stream wake-up:
            // profiles/r05_small_step_timeline.txt); a step in one part keeps them on the device and sends them home in one
            // contiguous run behind its last kernel
            uint32_t* const counts_to = (epilogue && ways > 1 && direct) ? count_view : o.out_count.p;
            rc = enqueue_compute(s, v, ow, true, s->exact_always ? MODE_EXACT : MODE_SCORE, st, rec, b->maybe_wide, start, counts_to, 0,
                                 defer_retry ? 1 : 0, b->widest);
            if (rc != SAGE_HIP_OK) {
                for (uint32_t k = 0; k < wy; k++) (void)hipStreamSynchronize(k ? s->way_stream[k - 1] : s->stream);
                return rc;
            }
            if (epilogue) {  // this part's PSM counts and counters in one launch, straight into the page-locked destinations
                EpilogueParts parts{};
                parts.n = 1;
                parts.src[0] = ow.counters.p;
                parts.dst[0] = ow.h_counters_view;
                const bool counts_home = counts_to == count_view;  // (the kernels stored them there already)
                launch_epilogue(o.out_count.p, counts_home ? 0u : v.n, count_view, ways > 1 ? v.order : nullptr, parts, st);
                HIP_TRY(hipGetLastError());
            } else {
                HIP_TRY(hipMemcpyAsync(ow.h_counters, ow.counters.p, 2 * CTR_COUNT * 4, hipMemcpyDeviceToHost, st));
            }
            // (every part's end: the parts run side by side and any of them may finish last — SageTiming::total_ms is the first
            // part's start to the LATEST end; round 5 recorded the last part's only and under-reported a step whose first part
            // outlasted it: ADVICE r05)
            if (s->timed) HIP_TRY(hipEventRecord(s->way_end[wy].e, st));
        }
        hipStream_t fin = ways > 1 ? s->way_stream[ways - 2] : s->stream;
        if (epilogue) {
            reset_done = true;
        } else if (b->n) {
            for (uint32_t wy = 0; wy + 1 < ways; wy++) HIP_TRY(wait_for_stream(wy ? s->way_stream[wy - 1] : s->stream));
            HIP_TRY(hipMemcpyAsync(out_count, o.out_count.p, (size_t)b->n * 4, hipMemcpyDeviceToHost, fin));
        }
        if (b->n && !direct) {
            for (uint32_t wy = 0; wy + 1 < ways; wy++) HIP_TRY(wait_for_stream(wy ? s->way_stream[wy - 1] : s->stream));
            HIP_TRY(hipMemcpyAsync(out, o.features.p, rec_bytes, hipMemcpyDeviceToHost, fin));
        }
        t_enq = us_since(t_enter);
        // the first parts first (they were launched first), the last one at the end
        for (uint32_t wy = 0; wy < ways; wy++) HIP_TRY(wait_for_stream(wy ? s->way_stream[wy - 1] : s->stream));
        t_wait = us_since(t_enter);
        if (reset_done)
            for (uint32_t wy = 0; wy < ways; wy++) s->outs[wy].counters_clean = true;
        bool redo = false;
        for (uint32_t wy = 0; wy < ways; wy++) {
            bool r = false;
            rc = collect(s, s->outs[wy], nullptr, &r);
            if (rc != SAGE_HIP_OK) return rc;
            redo = redo || r;
        }
        if (!redo && defer_retry) {
            bool any = false;
            for (uint32_t wy = 0; wy < ways; wy++) any = any || s->outs[wy].h_counters[CTR_RETRY] != 0;
            if (any) {  // the retry pass after all: the parts that have something to retry, each on its stream
                for (uint32_t wy = 0; wy < ways; wy++) {
                    OutSet& ow = s->outs[wy];
                    if (ow.h_counters[CTR_RETRY] == 0) {
                        ow.retry_deferred = false;
                        continue;
                    }
                    const uint32_t start = (uint32_t)((uint64_t)b->n * wy / ways), end = (uint32_t)((uint64_t)b->n * (wy + 1) / ways);
                    hipStream_t st = wy ? s->way_stream[wy - 1] : s->stream;
                    DevBatchView v = b->view;
                    v.order += start;
                    if (v.sched) v.sched += 2 * (size_t)start;
                    v.n = end - start;
                    uint32_t* const counts_to = (ways > 1) ? count_view : o.out_count.p;
                    rc = enqueue_compute(s, v, ow, true, MODE_SCORE, st, rec, false, start, counts_to, 0, 2);
                    if (rc != SAGE_HIP_OK) {
                        for (uint32_t k = 0; k < ways; k++) (void)hipStreamSynchronize(k ? s->way_stream[k - 1] : s->stream);
                        return rc;
                    }
                    EpilogueParts parts{};
                    parts.n = 1;
                    parts.src[0] = ow.counters.p;
                    parts.dst[0] = ow.h_counters_view;
                    launch_epilogue(o.out_count.p, counts_to == count_view ? 0u : v.n, count_view, nullptr, parts, st);
                    HIP_TRY(hipGetLastError());
                }
                for (uint32_t wy = 0; wy < ways; wy++) HIP_TRY(wait_for_stream(wy ? s->way_stream[wy - 1] : s->stream));
                for (uint32_t wy = 0; wy < ways; wy++) {
                    OutSet& ow = s->outs[wy];
                    ow.counters_clean = true;
                    if (!ow.retry_deferred) continue;
                    rc = collect_retry(s, ow);
                    if (rc != SAGE_HIP_OK) return rc;
                }
            } else {
                for (uint32_t wy = 0; wy < ways; wy++) s->outs[wy].retry_deferred = false;
            }
        }
        if (!redo) {
            s->retry_likely = s->timing.n_retry != 0;
            if (s->timed) {
                float wall = 0.f;
                for (uint32_t wy = 0; wy < ways; wy++) {
                    float w1 = 0.f;
                    HIP_TRY(hipEventElapsedTime(&w1, s->way_begin.e, s->way_end[wy].e));
                    wall = std::max(wall, w1);
                }
                s->timing.total_ms = wall;  // (prelim_ms / rescore_ms: summed over the parts, which overlap in time)
                s->kept_prelim_ms = s->timing.prelim_ms;
                s->kept_rescore_ms = s->timing.rescore_ms;
                s->kept_retry_ms = s->timing.retry_ms;
            } else {  // (an untimed step reports the kernel times of the last timed one; total_ms 0 says so)
                s->timing.prelim_ms = s->kept_prelim_ms;
                s->timing.rescore_ms = s->kept_rescore_ms;
                s->timing.retry_ms = s->kept_retry_ms;
                s->timing.total_ms = 0.f;
            }
            s->timing.n_ways = ways;
            return SAGE_HIP_OK;
        }
        if (attempt) return fail(SAGE_HIP_ERR_INTERNAL, "large-window queue not drained");
        reset_timing(s);
        b->maybe_wide = true;  // (and stays so: the batch holds large windows after all)
    }
}

int sage_hip_score_resident(SageScorer* s, SageDeviceBatch* b, SageFeature* out, uint32_t* out_count) {
    if (!s || !b || !out || !out_count) return fail(SAGE_HIP_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lock(s->mu);
    return score_resident_locked(s, b, out, out_count);
}

// Spectra [r0, r1) of a host batch through the three-stage pipeline: while chunk c is scored on the compute stream, chunk
// c + 1 is staged and uploaded on the copy stream and the PSM records of chunk c - 1 return on the download stream.  Mirrors the
// reference's reader -> processor -> search overlap (runner.rs:365-375, 450-461) at the PCIe boundary.
// `pending` (may be null): the window estimate is still being worked out on a helper thread — the first chunk is staged and its
// upload enqueued without it (neither needs it), then it is waited for and written into the chunk's view (`est` is updated).
static int score_range(SageScorer* s, const SageSpectrumBatch* b, uint32_t r0, uint32_t r1, SageFeature* out, uint32_t* out_count,
                       uint32_t chunk, bool peaks_locked, WindowEstimate& est, std::vector<std::pair<uint32_t, uint32_t>>& overflowed,
                       std::future<WindowEstimate>* pending = nullptr) {
    const uint32_t rp = s->params.report_psms;
    // page-locked result arrays (sage_hip_host_alloc) receive the records from the kernels' own stores; pageable ones through a
    // page-locked landing block per slot (an asynchronous copy into pageable memory would stall the pipeline)
    const bool out_locked = is_page_locked(out) && is_page_locked(out_count);
    SageFeature* const direct = s->zero_copy && out_locked ? (SageFeature*)device_view(out) : nullptr;
    const int mode = s->exact_always ? MODE_EXACT : MODE_SCORE;
    struct Pending {
        uint32_t c0, c1;
        int slot;
    };
    constexpr int NSLOT = 4;
    Pending pend[NSLOT];
    bool has[NSLOT] = {false, false, false, false};
    // SAGE_HIP_TIMING=1: host-side wall clock of the pipeline's steps on stderr (microseconds since the call began)
    static const bool trace_on = std::getenv("SAGE_HIP_TIMING") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto trace = [&](const char* what, int k_) {
        if (trace_on)
            fprintf(stderr, "[sage_hip] score_range %8.1f us  %s %d\n",
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count(), what, k_);
    };
    // an error with chunks still in flight: nothing may keep writing into the caller's arrays after the call has returned
    auto bail = [&](int rc) -> int {
        (void)hipStreamSynchronize(s->up_stream);
        (void)hipStreamSynchronize(s->stream);
        (void)hipStreamSynchronize(s->way_stream[0]);
        (void)hipStreamSynchronize(s->down_stream);
        for (OutSet& o : s->outs) o.in_flight = false;
        return rc;
    };
#define HIP_TRY_BAIL(expr)                                                                  \
    do {                                                                                    \
        const hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) return bail(fail(SAGE_HIP_ERR_HIP, hipGetErrorString(e_)));   \
    } while (0)
    // kernels + the way home of the records of one chunk (its input is resident in s->slots[slot])
    // Two chunks of a batch WITHOUT large windows are scored side by side, chunk k on compute lane k & 1 (own stream, own working
    // set): the cold start, the tail and the retry chain of one chunk's kernels are filled by the other's (what two scorer
    // handles on two host threads do, §6 of DESIGN.md, here inside the pipeline).  With large windows: lane 0 only.
    auto lane_of = [&](int slot, bool wide) { return (!wide && mode == MODE_SCORE && s->two_lanes) ? (slot & 1) : 0; };
    auto launch = [&](int slot, uint32_t c0, uint32_t c1, bool wide) -> int {
        SageDeviceBatch& in = s->slots[slot];
        OutSet& o = s->outs[slot];
        const int lane = lane_of(slot, wide);
        hipStream_t cs = lane ? s->way_stream[0] : s->stream;
        HIP_TRY_BAIL(hipStreamWaitEvent(cs, in.up_done.e, 0));
        int rc = enqueue_compute(s, in.view, o, true, mode, cs, direct ? direct + (size_t)c0 * rp : nullptr, wide, 0, nullptr, lane);
        if (rc != SAGE_HIP_OK) return bail(rc);
        HIP_TRY_BAIL(hipEventRecord(o.comp_done.e, cs));
        // the next upload into this slot (chunk k + 4) is enqueued only after finish(slot) has waited for this chunk's
        // download, which itself follows its kernels: no device-side guard is needed for the input buffers
        HIP_TRY_BAIL(hipStreamWaitEvent(s->down_stream, o.comp_done.e, 0));
        HIP_TRY_BAIL(hipMemcpyAsync(o.h_counters, o.counters.p, 2 * CTR_COUNT * 4, hipMemcpyDeviceToHost, s->down_stream));
        SageFeature* dst_f = out + (size_t)c0 * rp;
        uint32_t* dst_c = out_count + c0;
        if (!out_locked) {
            const size_t fb = (((size_t)(c1 - c0) * rp * sizeof(SageFeature) + 63) / 64) * 64;
            const hipError_t e_ = o.h_out.reserve(fb + (size_t)(c1 - c0) * 4);
            if (e_ != hipSuccess) return bail(fail(SAGE_HIP_ERR_HIP, hipGetErrorString(e_)));
            dst_f = (SageFeature*)o.h_out.p;
            dst_c = (uint32_t*)(o.h_out.p + fb);
        }
        HIP_TRY_BAIL(hipMemcpyAsync(dst_c, o.out_count.p, (size_t)(c1 - c0) * 4, hipMemcpyDeviceToHost, s->down_stream));
        if (!direct)
            HIP_TRY_BAIL(hipMemcpyAsync(dst_f, o.features.p, (size_t)(c1 - c0) * rp * sizeof(SageFeature), hipMemcpyDeviceToHost, s->down_stream));
        HIP_TRY_BAIL(hipEventRecord(o.down_done.e, s->down_stream));
        return SAGE_HIP_OK;
    };
    auto finish = [&](int slot) -> int {
        if (!has[slot]) return SAGE_HIP_OK;
        has[slot] = false;
        OutSet& o = s->outs[slot];
        HIP_TRY_BAIL(hipEventSynchronize(o.down_done.e));
        bool ovf = false, redo = false;
        int rc = collect(s, o, &ovf, &redo);
        if (rc != SAGE_HIP_OK) return bail(rc);
        if (redo) {
            // the chunk held large windows although the sample said otherwise: once more with the large-window kernels (its
            // input is still resident; the working set is shared, so behind whatever the compute stream is doing now)
            est.maybe_wide = true;
            rc = launch(slot, pend[slot].c0, pend[slot].c1, true);
            if (rc != SAGE_HIP_OK) return rc;
            HIP_TRY_BAIL(hipEventSynchronize(o.down_done.e));
            rc = collect(s, o, &ovf, nullptr);
            if (rc != SAGE_HIP_OK) return bail(rc);
        }
        if (!out_locked) {
            const uint32_t c0 = pend[slot].c0, cn = pend[slot].c1 - pend[slot].c0;
            std::memcpy(out + (size_t)c0 * rp, o.h_out.p, (size_t)cn * rp * sizeof(SageFeature));
            std::memcpy(out_count + c0, o.h_out.p + (((size_t)cn * rp * sizeof(SageFeature) + 63) / 64) * 64, (size_t)cn * 4);
        }
        if (ovf) overflowed.push_back({pend[slot].c0, pend[slot].c1});
        return SAGE_HIP_OK;
    };
    // (pieces of equal size: short first and last pieces — the kernels start earlier, less is left when the link goes quiet — were
    // measured and lose, round 3: C3 44.5 against 46.9 M spectra/s; round 4, the last piece alone cut in 1/2, 1/4, 1/4 on the
    // faster kernels and with the schedule sort off the link: 46 against 54 M spectra/s)
    int k = 0;
    for (uint32_t c0 = r0; c0 < r1; c0 += chunk, k++) {
        const uint32_t c1 = (uint32_t)std::min<uint64_t>((uint64_t)c0 + chunk, r1);
        const int slot = k % NSLOT;
        trace("wait for slot of chunk", k);
        int rc = finish(slot);  // chunk k - 4 used this slot: its records are home, its buffers are free
        if (rc != SAGE_HIP_OK) return rc;
        trace("stage chunk", k);
        SageDeviceBatch& in = s->slots[slot];
        // (the uploads of chunk k - 4 finished long ago — its kernels ran — so the staging block may be rewritten)
        rc = stage_and_upload(s, &in, b, c0, c1, peaks_locked, est, s->up_stream);
        if (pending && pending->valid()) {  // (also on the error path: the helper thread reads the caller's arrays)
            est = pending->get();
            in.view.probe = est.probe;
            in.maybe_wide = est.maybe_wide;
            trace("window estimate joined", k);
        }
        if (rc != SAGE_HIP_OK) return bail(rc);
        trace("launch chunk", k);
        rc = launch(slot, c0, c1, est.maybe_wide);
        if (rc != SAGE_HIP_OK) return rc;
        trace("launched chunk", k);
        pend[slot] = Pending{c0, c1, slot};
        has[slot] = true;
    }
#undef HIP_TRY_BAIL
    // a working set (candidate lists, arena) is shared by the chunks of its lane: their kernels run in order on the lane's
    // stream, and a chunk's records leave through its own OutSet, so the next chunk's kernels may start while they are on the way
    trace("drain", k);
    for (int j = 0; j < NSLOT; j++) {  // oldest first
        const int rc = finish((k + j) % NSLOT);
        if (rc != SAGE_HIP_OK) return rc;
    }
    trace("done", k);
    return SAGE_HIP_OK;
}

int sage_hip_score_batch(SageScorer* s, const SageSpectrumBatch* b, SageFeature* out, uint32_t* out_count) {
    if (!s || !b || (b->n_spectra && (!out || !out_count))) return fail(SAGE_HIP_ERR_INVALID, "null argument");
    int rc = check_batch_args(b);
    if (rc != SAGE_HIP_OK) return rc;
    std::lock_guard<std::mutex> lock(s->mu);
    HIP_TRY(hipSetDevice(s->db->device));
    reset_timing(s);
    const uint32_t n = b->n_spectra;
    if (n == 0) return SAGE_HIP_OK;
    const auto t_call = std::chrono::steady_clock::now();
    const bool peaks_locked = is_page_locked(b->masses) && is_page_locked(b->intensities);
    // The window estimate (1 024 sampled spectra x two searches over the peptide masses: ~0.5 ms of cache misses on a 4.75 M
    // peptide list) runs on a helper thread while this one stages the first chunk and enqueues its upload, which need neither
    // of its answers; score_range joins it before the chunk's first launch.
    std::future<WindowEstimate> pending = std::async(std::launch::async, [s, n, b]() {
        return choose_probe(s, n, b->precursor_mz, b->precursor_charge, b->isolation_lo, b->isolation_hi);
    });
    WindowEstimate est{};
    est.probe = 1;
    est.maybe_wide = false;
    (void)t_call;
    std::vector<std::pair<uint32_t, uint32_t>> todo, next;
    rc = score_range(s, b, 0, n, out, out_count, s->chunk, peaks_locked, est, todo, &pending);
    if (pending.valid()) est = pending.get();  // (score_range left before its first chunk: an error)
    if (rc != SAGE_HIP_OK) return rc;
    // chunks whose large-window candidates did not fit the arena: again in halves (the arena is sized for the chunk, so a
    // piece with the same arena and half the spectra has twice the room per spectrum)
    while (!todo.empty()) {
        next.clear();
        for (auto& r : todo) {
            const uint32_t len = r.second - r.first;
            if (len <= 1)
                return fail(SAGE_HIP_ERR_UNSUPPORTED, "large-window candidate arena exhausted by a single spectrum (" +
                                                          std::to_string(s->ws.arena.n >> 18) + " MiB): raise SAGE_HIP_ARENA_MB");
            rc = score_range(s, b, r.first, r.second, out, out_count, (len + 1) / 2, peaks_locked, est, next);
            if (rc != SAGE_HIP_OK) return rc;
        }
        todo.swap(next);
    }
    return SAGE_HIP_OK;
}

int sage_hip_annotate_resident(SageScorer* s, SageDeviceBatch* b, const SageFeature* features, const uint32_t* counts,
                               SageFragments* out) {
    if (!s || !b || !features || !counts || !out || !out->psm_off) return fail(SAGE_HIP_ERR_INVALID, "null argument");
    if (b->device != s->db->device) return fail(SAGE_HIP_ERR_INVALID, "batch and scorer live on different devices");
    std::lock_guard<std::mutex> lock(s->mu);
    HIP_TRY(hipSetDevice(s->db->device));
    const uint32_t n = b->n, rp = s->params.report_psms;
    const size_t slots = (size_t)n * rp;
    // a PSM's Fragments hold exactly matched_b + matched_y entries (scoring.rs:725-752) == Feature.matched_peaks
    uint64_t total = 0;
    for (uint32_t i = 0; i < n; i++)
        for (uint32_t r = 0; r < rp; r++) {
            out->psm_off[(size_t)i * rp + r] = total;
            if (r < counts[i]) {
                if (features[(size_t)i * rp + r].peptide_idx >= s->db->view.np) return fail(SAGE_HIP_ERR_INVALID, "feature peptide_idx out of range");
                total += features[(size_t)i * rp + r].matched_peaks;
            }
        }
    out->psm_off[slots] = total;
    if (total > out->capacity) return fail(SAGE_HIP_ERR_INVALID, "SageFragments.capacity is smaller than the sum of matched_peaks");
    if (total && (!out->kinds || !out->charges || !out->fragment_ordinals || !out->intensities || !out->mz_calculated || !out->mz_experimental))
        return fail(SAGE_HIP_ERR_INVALID, "missing SageFragments arrays");
    if (n == 0 || total == 0) return SAGE_HIP_OK;
    HIP_TRY(s->an_feats.reserve(slots));
    HIP_TRY(s->an_counts.reserve(n));
    HIP_TRY(s->an_off.reserve(slots + 1));
    HIP_TRY(s->an_kinds.reserve(total));
    HIP_TRY(s->an_charges.reserve(total));
    HIP_TRY(s->an_ord.reserve(total));
    HIP_TRY(s->an_int.reserve(total));
    HIP_TRY(s->an_calc.reserve(total));
    HIP_TRY(s->an_exp.reserve(total));
    hipStream_t st = s->stream;
    HIP_TRY(hipMemcpyAsync(s->an_feats.p, features, slots * sizeof(SageFeature), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->an_counts.p, counts, (size_t)n * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->an_off.p, out->psm_off, (slots + 1) * 8, hipMemcpyHostToDevice, st));
    DevFragments df{total, s->an_kinds.p, s->an_charges.p, s->an_ord.p, s->an_int.p, s->an_calc.p, s->an_exp.p};
    launch_annotate(s->db->view, s->dev, b->view, s->an_feats.p, s->an_counts.p, s->an_off.p, df, st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out->kinds, s->an_kinds.p, total, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out->charges, s->an_charges.p, total * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out->fragment_ordinals, s->an_ord.p, total * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out->intensities, s->an_int.p, total * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out->mz_calculated, s->an_calc.p, total * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out->mz_experimental, s->an_exp.p, total * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return SAGE_HIP_OK;
}

int sage_hip_quick_score_resident(SageScorer* s, SageDeviceBatch* b, int prefilter_low_memory, uint8_t* keep) {
    if (!s || !b || !keep) return fail(SAGE_HIP_ERR_INVALID, "null argument");
    if (b->device != s->db->device) return fail(SAGE_HIP_ERR_INVALID, "batch and scorer live on different devices");
    std::lock_guard<std::mutex> lock(s->mu);
    HIP_TRY(hipSetDevice(s->db->device));
    reset_timing(s);
    OutSet& o = s->outs[0];
    int rc = enqueue_compute(s, b->view, o, false, MODE_FAST, s->stream);  // Scorer::initial_hits
    if (rc != SAGE_HIP_OK) return rc;
    const uint64_t np = s->db->view.np;
    HIP_TRY(s->keep.reserve(np));
    HIP_TRY(hipMemsetAsync(s->keep.p, 0, std::max<uint64_t>(np, 1), s->stream));
    DevWork w = make_work(s, o, 0);
    if (s->ws.hugebuf.p && o.huge) {  // (enqueue_compute above decided: the wide-list kernels' arrays live in the global workspace)
        w.hugebuf = s->ws.hugebuf.p;
        w.huge_stride = (uint32_t)huge_stride_bytes(s->dev);
    }
    if (prefilter_low_memory)
        launch_rescore(s->db->view, s->dev, b->view, w, s->lnfact.p, (uint32_t)s->lnfact.n, s->db->max_ions, o.features.p,
                       o.out_count.p, s->keep.p, s->stream);
    else
        launch_quick_mark(s->dev, b->view, w, s->keep.p, s->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(o.h_counters, o.counters.p, 2 * CTR_COUNT * 4, hipMemcpyDeviceToHost, s->stream));
    std::vector<uint8_t> h(np);
    HIP_TRY(hipMemcpyAsync(h.data(), s->keep.p, np, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    rc = collect(s, o, nullptr);
    if (rc != SAGE_HIP_OK) return rc;
    for (uint64_t i = 0; i < np; i++) keep[i] |= h[i];
    return SAGE_HIP_OK;
}

int sage_hip_initial_hits(SageScorer* s, SageDeviceBatch* b, uint64_t* packed, uint32_t cap, uint32_t* len,
                          uint64_t* matched_peaks, uint64_t* scored_candidates) {
    if (!s || !b || !packed || !len) return fail(SAGE_HIP_ERR_INVALID, "null argument");
    if (cap < s->dev.kmax) return fail(SAGE_HIP_ERR_INVALID, "cap must be >= max(50, 2*report_psms)");
    if (b->device != s->db->device) return fail(SAGE_HIP_ERR_INVALID, "batch and scorer live on different devices");
    std::lock_guard<std::mutex> lock(s->mu);
    HIP_TRY(hipSetDevice(s->db->device));
    reset_timing(s);
    OutSet& o = s->outs[0];
    int rc = enqueue_compute(s, b->view, o, false, MODE_EXACT, s->stream);  // the reference's heap layouts
    if (rc != SAGE_HIP_OK) return rc;
    HIP_TRY(hipMemcpyAsync(o.h_counters, o.counters.p, 2 * CTR_COUNT * 4, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    rc = collect(s, o, nullptr);
    if (rc != SAGE_HIP_OK) return rc;
    const uint32_t n = b->n, kmax = s->dev.kmax;
    std::vector<uint64_t> c((size_t)n * kmax);
    std::vector<uint32_t> tot((size_t)n * 2), st(n);
    HIP_TRY(hipMemcpy(c.data(), s->ws.cand.p, c.size() * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(len, s->ws.cand_len.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(tot.data(), s->ws.totals.p, tot.size() * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(st.data(), s->ws.status.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; i++) {
        if (st[i] != ST_OK && st[i] != ST_OK_ORDERED) return fail(SAGE_HIP_ERR_UNSUPPORTED, "spectrum " + std::to_string(i) + ": status " + std::to_string(st[i]));
        for (uint32_t j = 0; j < len[i]; j++) packed[(size_t)i * cap + j] = c[(size_t)i * kmax + j];
        if (matched_peaks) matched_peaks[i] = tot[2 * i];
        if (scored_candidates) scored_candidates[i] = tot[2 * i + 1];
    }
    return SAGE_HIP_OK;
}

// debugging aid (not part of the drop-in surface): cumulative per-phase shader cycles, [4 kernels][8 phases]
int sage_hip_debug_phase_cycles(SageScorer* s, unsigned long long* out32) {
    if (!s || !out32) return fail(SAGE_HIP_ERR_INVALID, "null argument");
    if (!s->dbg.p) return fail(SAGE_HIP_ERR_INVALID, "set SAGE_HIP_PHASE_CLOCKS=1 before creating the scorer");
    std::vector<unsigned long long> all(4096 * 32);
    HIP_TRY(hipMemcpy(all.data(), s->dbg.p, all.size() * 8, hipMemcpyDeviceToHost));
    for (int k = 0; k < 32; k++) out32[k] = 0;
    for (size_t b = 0; b < 4096; b++)
        for (int k = 0; k < 32; k++) out32[k] += all[b * 32 + k];
    return SAGE_HIP_OK;
}

int sage_hip_host_alloc(uint64_t bytes, void** out) {
    if (!out) return fail(SAGE_HIP_ERR_INVALID, "null argument");
    HIP_TRY(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    if (void* d = device_view(*out)) {  // (device_view_cached's registry)
        std::lock_guard<std::mutex> lock(g_blocks_mu);
        g_blocks.push_back(HostBlock{(const unsigned char*)*out, (unsigned char*)d, (size_t)(bytes ? bytes : 1)});
    }
    return SAGE_HIP_OK;
}
void sage_hip_host_free(void* p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lock(g_blocks_mu);
        for (size_t i = 0; i < g_blocks.size(); i++)
            if (g_blocks[i].host == (const unsigned char*)p) {
                g_blocks.erase(g_blocks.begin() + (long)i);
                break;
            }
    }
    (void)hipHostFree(p);
}

int sage_hip_scorer_set_timing_interval(SageScorer* s, uint32_t every) {
    if (!s) return fail(SAGE_HIP_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lock(s->mu);
    s->timing_every = every;
    s->timing_calls = 0;
    return SAGE_HIP_OK;
}

int sage_hip_last_timing(const SageScorer* s, SageTiming* out) {
    if (!s || !out) return fail(SAGE_HIP_ERR_INVALID, "null argument");
    *out = s->timing;
    return SAGE_HIP_OK;
}

}  // extern "C"
