// capi.hip — the C ABI declared in include/sage_hip.h.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

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
    size_t bytes() const { return n * sizeof(T); }
};

}  // namespace

struct SageHostDb {
    HostDb db;
};

struct SageDeviceDb {
    int device = 0;
    DevBuf<float> pep_mono;
    DevBuf<float> ions;
    DevBuf<uint64_t> ion_off;
    DevBuf<uint32_t> pep_info;
    DevBuf<SageTheoretical> pm_frag;  // peptide-major copy (narrow kernel, small windows)
    DevBuf<uint64_t> pm_off;
    std::vector<float> h_pep_mono;    // host copy: window-size estimate at batch upload
    DevBuf<SageTheoretical> tm_frag;  // tile-major copy + position table for the large-window kernel
    DevBuf<uint32_t> tm_lut;
    DevBuf<SageTheoretical> tm2_frag; // small-tile copy + table for the narrow kernel's per-peak lookups
    DevBuf<uint32_t> tm2_lut;
    uint32_t max_ions = 0;
    DevDbView view{};
    uint64_t bytes = 0;
};

struct SageScorer {
    SageDeviceDb* db = nullptr;
    SageScorerParams params{};
    DevScorer dev{};
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {};
    DevBuf<double> lnfact;
    DevBuf<unsigned long long> dbg;  // SAGE_HIP_PHASE_CLOCKS=1: per-phase cycle accumulators
    uint32_t tile_blocks = 0;        // persistent workgroups of the large-window kernel
    SageTiming timing{};
    // per-batch work buffers, grown on demand
    DevBuf<uint64_t> cand;
    DevBuf<uint32_t> cand_len, totals, status, n_deferred, out_count, queue, retry;
    bool exact_always = false;  // SAGE_HIP_EXACT=1: never use the order-free trims
    // large-window pipeline scratch (device_types.h: DevWork)
    DevBuf<QueryRec> qrec;
    DevBuf<uint16_t> seeds;
    DevBuf<uint64_t> qres;
    DevBuf<uint32_t> arena;
    DevBuf<TileParams> tile_params;
    uint32_t qmax = 1;
    DevBuf<SageFeature> features;
    uint32_t work_n = 0;
    uint32_t* h_counters = nullptr;  // pinned [2]: deferred, overflow
};

struct SageDeviceBatch {
    int device = 0;
    uint32_t n = 0;
    DevBuf<uint64_t> peak_off;
    DevBuf<float> masses, intensities, precursor_mz, iso_lo, iso_hi, tic, rt, ims;
    DevBuf<uint8_t> charge;
    DevBuf<uint32_t> file_id, order;
    DevBatchView view{};
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
    uint32_t max_ions = 0;
    for (uint64_t i = 0; i < np; i++) {
        const uint64_t len = v->seq_off[i + 1] - v->seq_off[i];
        if (len > 0xFFFF) return fail(SAGE_HIP_ERR_UNSUPPORTED, "peptide longer than 65535 residues");
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
        HIP_TRY(d->ions.alloc(ion_off[np]));
        HIP_TRY(d->pm_frag.alloc(nf));
        HIP_TRY(d->tm_frag.alloc(nf + 2));
        hipError_t be = (hipError_t)generate_fragments_on_device(np, nk, d_kinds.p, d_seq_off.p, d_seq.p, d_mods.p, d_nterm.p, d->pep_mono.p,
                                                                v->min_ion_index, d->ion_off.p, d->pm_off.p, d->ions.p, d->pm_frag.p, nullptr);
        uint32_t* lut_p = nullptr;
        if (be == hipSuccess)
            be = (hipError_t)build_tile_copy_on_device(d->pm_frag.p, nf, tile_shift, (uint32_t)n_tiles, d_tile_off.p, lut_scale,
                                                       d->tm_frag.p, &lut_p, &lut_stride, nullptr);
        if (be != hipSuccess)
            return fail(be == hipErrorOutOfMemory ? SAGE_HIP_ERR_OOM : SAGE_HIP_ERR_HIP, std::string("device index build: ") + hipGetErrorString(be));
        d->tm_lut.p = lut_p;
        d->tm_lut.n = (size_t)n_tiles * lut_stride;
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
        std::vector<float> ions(ion_off[np]);
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
        std::vector<uint32_t> lut((size_t)n_tiles * lut_stride);
        parallel_for(n_tiles, 1, [&](size_t tb, size_t te, unsigned) {
            for (size_t t = tb; t < te; t++) {
                uint64_t pos = tile_off[t];
                const uint64_t tend = tile_off[t + 1];
                uint32_t* row = lut.data() + t * lut_stride;
                for (uint32_t c = 0; c < lut_stride; c++) {
                    const double edge = (double)c / (double)lut_scale;
                    // NaN and m/z beyond the table (non-finite or > 250 kDa) compare false and stay in the last cell's run
                    while (pos < tend && (double)tm[pos].fragment_mz < edge) pos++;
                    row[c] = (uint32_t)pos;
                }
                row[0] = (uint32_t)tile_off[t];  // a window starting below cell 0 starts at the tile's first entry
                row[lut_stride - 1] = (uint32_t)tend;
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
        uint32_t tile2_shift = 12;
        if (const char* e = getenv("SAGE_HIP_TILE2_SHIFT")) tile2_shift = (uint32_t)std::min(16, std::max(6, atoi(e)));
        const float lut2_scale = 32.0f;  // (a power of two, like lut_scale)
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
        d->tm2_lut.p = lut2_p;
        d->tm2_lut.n = (size_t)n_tiles2 * lut2_stride;
        d->view.tm2_frag = d->tm2_frag.p;
        d->view.tm2_lut = d->tm2_lut.p;
        d->view.tile2_shift = tile2_shift;
        d->view.n_tiles2 = (uint32_t)n_tiles2;
        d->view.lut2_stride = lut2_stride;
        d->view.lut2_scale = lut2_scale;
    }
    d->max_ions = max_ions;
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
    d->bytes = d->pep_mono.bytes() + d->pm_frag.bytes() + d->pm_off.bytes() + d->ions.bytes() + d->ion_off.bytes() +
               d->pep_info.bytes() + d->tm_frag.bytes() + d->tm_lut.bytes() + d->tm2_frag.bytes() + d->tm2_lut.bytes();
    *out = d.release();
    return SAGE_HIP_OK;
}

void sage_hip_db_destroy(SageDeviceDb* db) {
    if (!db) return;
    (void)hipSetDevice(db->device);
    delete db;
}
uint64_t sage_hip_db_device_bytes(const SageDeviceDb* db) { return db ? db->bytes : 0; }

int sage_hip_scorer_create(SageDeviceDb* db, const SageScorerParams* p, SageScorer** out) {
    if (!db || !p || !out) return fail(SAGE_HIP_ERR_INVALID, "null argument");
    if (p->report_psms == 0) return fail(SAGE_HIP_ERR_INVALID, "report_psms must be >= 1");
    if (p->report_psms > 32) return fail(SAGE_HIP_ERR_UNSUPPORTED, "report_psms > 32 (k-select wider than one wavefront)");
    if (p->min_isotope_err > p->max_isotope_err) return fail(SAGE_HIP_ERR_INVALID, "min_isotope_err > max_isotope_err");
    if (p->min_precursor_charge > p->max_precursor_charge || p->min_precursor_charge == 0)
        return fail(SAGE_HIP_ERR_INVALID, "precursor charge range must be [lo >= 1, hi >= lo]");
    if (p->score_type != 0 && p->score_type != 1) return fail(SAGE_HIP_ERR_INVALID, "unknown score_type");
    HIP_TRY(hipSetDevice(db->device));
    auto s = std::make_unique<SageScorer>();
    s->db = db;
    s->params = *p;
    DevScorer& d = s->dev;
    d.precursor_tol = {p->precursor_tol.kind, p->precursor_tol.lo, p->precursor_tol.hi};
    d.fragment_tol = {p->fragment_tol.kind, p->fragment_tol.lo, p->fragment_tol.hi};
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
    if (const char* e = getenv("SAGE_HIP_EXACT")) s->exact_always = atoi(e) != 0;
    if (const char* e = getenv("SAGE_HIP_WCAP")) d.wcap = (uint32_t)std::min(16384, std::max(64, atoi(e)));  // (32-bit heap keys need <= 65536)
    HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    for (auto& e : s->ev) HIP_TRY(hipEventCreate(&e));
    // lnfact (scoring.rs:170-177) tabulated with the host libm so the factorial terms are bit-identical
    // to a CPU evaluation
    std::vector<double> tbl(4096);
    tbl[0] = 1.0;
    for (uint32_t n = 1; n < tbl.size(); n++) {
        const double x = (double)n;
        tbl[n] = x * std::log(x) - x + 0.5 * std::log(x) + 0.5 * std::log(M_PI * 2.0 * x);
    }
    HIP_TRY(s->lnfact.upload(tbl.data(), tbl.size()));
    // large-window kernel: persistent workgroups, as many as the LDS tiles allow to be resident
    {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, db->device));
        const size_t tile_lds = ((size_t)1 << db->view.tile_shift) * 2 + 12 * 1024;  // counters + bitmap + windows (typical)
        const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / tile_lds));
        s->tile_blocks = (uint32_t)prop.multiProcessorCount * per_cu;
        if (const char* e = getenv("SAGE_HIP_TILE_BLOCKS")) s->tile_blocks = (uint32_t)std::max(1, atoi(e));
        HIP_TRY((hipError_t)tile_kernel_prepare(160 * 1024));
    }
    s->qmax = queries_per_spectrum(d);
    HIP_TRY(s->n_deferred.alloc(CTR_COUNT));
    HIP_TRY(s->tile_params.alloc(1));
    HIP_TRY(hipHostMalloc((void**)&s->h_counters, CTR_COUNT * 4, hipHostMallocDefault));
    if (const char* e = getenv("SAGE_HIP_PHASE_CLOCKS")) {
        if (atoi(e) > 0) {
            HIP_TRY(s->dbg.alloc(4096 * 32));
            HIP_TRY(hipMemset(s->dbg.p, 0, 4096 * 32 * 8));
        }
    }
    *out = s.release();
    return SAGE_HIP_OK;
}

void sage_hip_scorer_destroy(SageScorer* s) {
    if (!s) return;
    (void)hipSetDevice(s->db->device);
    for (auto& e : s->ev)
        if (e) (void)hipEventDestroy(e);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    if (s->h_counters) (void)hipHostFree(s->h_counters);
    delete s;
}

// precursor-side arrays, launch schedule, kernel variant and the view of a batch whose peak arrays are already on the device
static int finish_batch(SageScorer* s, SageDeviceBatch* d, uint32_t n, const float* precursor_mz, const uint8_t* precursor_charge,
                        const float* isolation_lo, const float* isolation_hi, const float* scan_start_time,
                        const float* inverse_ion_mobility, const uint32_t* file_id, uint32_t pcap) {
    uint32_t zmax = 0;
    bool any_unknown = false;
    for (uint32_t i = 0; i < n; i++) {
        zmax = std::max<uint32_t>(zmax, precursor_charge[i]);
        any_unknown = any_unknown || precursor_charge[i] == 0;
    }
    // largest fragment charge any spectrum of this batch can ask for (scoring.rs:239-247)
    uint32_t fzcap = 1;
    const SageScorerParams& p = s->params;
    const bool ranged = p.wide_window || p.override_precursor_charge || any_unknown;
    for (uint32_t z = 1; z <= 255; z++) {
        const bool used = (ranged && z >= p.min_precursor_charge && z <= p.max_precursor_charge) ||
                          (!p.wide_window && !p.override_precursor_charge && z <= zmax);
        if (used) fzcap = std::max(fzcap, sagecore::max_fragment_charge(p.max_fragment_charge, z) - 1);
    }
    // schedule spectra by ascending neutral precursor mass: wavefronts resident together then read
    // overlapping ranges of the index and of the ion table (outputs keep input order)
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; i++) order[i] = i;
    {
        std::vector<float> key(n);
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t z = precursor_charge[i] ? precursor_charge[i] : p.min_precursor_charge;
            key[i] = (precursor_mz[i] - sagecore::PROTON) * (float)z;
        }
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t c) { return key[a] < key[c]; });
    }
    // narrow-kernel variant for this batch: mean candidate-window size of (a sample of) its spectra, first query each
    {
        const std::vector<float>& pm = s->db->h_pep_mono;
        const uint32_t step = std::max<uint32_t>(1, n / 2048);
        double sum = 0.0;
        uint32_t cnt = 0;
        for (uint32_t i = 0; i < n; i += step, cnt++) {
            const uint32_t z = precursor_charge[i] ? precursor_charge[i] : p.min_precursor_charge;
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
            sum += (double)(std::upper_bound(pm.begin(), pm.end(), hi) - std::lower_bound(pm.begin(), pm.end(), lo));
        }
        const double mean_window = cnt ? sum / cnt : 0.0;
        d->view.probe = mean_window > 96.0 ? 1u : 0u;
        if (const char* e = getenv("SAGE_HIP_NARROW")) d->view.probe = std::string(e) == "probe" ? 1u : std::string(e) == "stream" ? 0u : d->view.probe;
    }
    HIP_TRY(d->order.upload(order.data(), n));
    HIP_TRY(d->precursor_mz.upload(precursor_mz, n));
    HIP_TRY(d->charge.upload(precursor_charge, n));
    if (isolation_lo && isolation_hi) {
        HIP_TRY(d->iso_lo.upload(isolation_lo, n));
        HIP_TRY(d->iso_hi.upload(isolation_hi, n));
    }
    if (scan_start_time) HIP_TRY(d->rt.upload(scan_start_time, n));
    if (inverse_ion_mobility) HIP_TRY(d->ims.upload(inverse_ion_mobility, n));
    if (file_id) HIP_TRY(d->file_id.upload(file_id, n));
    DevBatchView& v = d->view;
    v.n = n;
    v.peak_off = d->peak_off.p;
    v.masses = d->masses.p;
    v.intensities = d->intensities.p;
    v.precursor_mz = d->precursor_mz.p;
    v.precursor_charge = d->charge.p;
    v.isolation_lo = d->iso_lo.p;
    v.isolation_hi = d->iso_hi.p;
    v.tic = d->tic.p;
    v.rt = d->rt.p;
    v.ims = d->ims.p;
    v.file_id = d->file_id.p;
    v.order = d->order.p;
    v.pcap = pcap;
    v.fzcap = fzcap;
    return SAGE_HIP_OK;
}

int sage_hip_batch_upload(SageScorer* s, const SageSpectrumBatch* b, SageDeviceBatch** out) {
    if (!s || !b || !out) return fail(SAGE_HIP_ERR_INVALID, "null argument");
    if (b->n_spectra && (!b->peak_off || !b->precursor_mz || !b->precursor_charge || !b->total_ion_current))
        return fail(SAGE_HIP_ERR_INVALID, "missing required spectrum arrays");
    HIP_TRY(hipSetDevice(s->db->device));
    auto d = std::make_unique<SageDeviceBatch>();
    d->device = s->db->device;
    const uint32_t n = b->n_spectra;
    d->n = n;
    const uint64_t total = n ? b->peak_off[n] : 0;
    uint32_t pcap = 1;
    for (uint32_t i = 0; i < n; i++) {
        if (b->peak_off[i + 1] < b->peak_off[i]) return fail(SAGE_HIP_ERR_INVALID, "peak_off is not monotone");
        pcap = std::max<uint32_t>(pcap, (uint32_t)(b->peak_off[i + 1] - b->peak_off[i]));
    }
    if (total && (!b->masses || !b->intensities)) return fail(SAGE_HIP_ERR_INVALID, "missing peak arrays");
    HIP_TRY(d->peak_off.upload(b->peak_off, n ? (size_t)n + 1 : 0));
    HIP_TRY(d->masses.upload(b->masses, total));
    HIP_TRY(d->intensities.upload(b->intensities, total));
    HIP_TRY(d->tic.upload(b->total_ion_current, n));
    int rc = finish_batch(s, d.get(), n, b->precursor_mz, b->precursor_charge, b->isolation_lo, b->isolation_hi,
                          b->scan_start_time, b->inverse_ion_mobility, b->file_id, pcap);
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
    HIP_TRY(hipSetDevice(s->db->device));
    auto d = std::make_unique<SageDeviceBatch>();
    d->device = s->db->device;
    d->n = n;
    const uint64_t total = n ? raw->peak_off[n] : 0;
    uint32_t rcap = 1;
    for (uint32_t i = 0; i < n; i++) {
        if (raw->peak_off[i + 1] < raw->peak_off[i]) return fail(SAGE_HIP_ERR_INVALID, "peak_off is not monotone");
        rcap = std::max<uint32_t>(rcap, (uint32_t)(raw->peak_off[i + 1] - raw->peak_off[i]));
    }
    if (total && (!raw->mz || !raw->intensities)) return fail(SAGE_HIP_ERR_INVALID, "missing peak arrays");
    uint32_t rpow2 = 1;
    while (rpow2 < rcap) rpow2 <<= 1;
    const size_t lds = process_lds_bytes(rcap, rpow2);
    if (lds > 160 * 1024)
        return fail(SAGE_HIP_ERR_UNSUPPORTED, "a spectrum has more raw peaks than the LDS staging holds (" + std::to_string(rcap) +
                                                  "): preprocess it with sage_hip_process_ms2");
    HIP_TRY((hipError_t)process_kernel_prepare(160 * 1024));
    const uint32_t stride = (uint32_t)std::min<uint64_t>(take_top_n, rcap);
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
    std::vector<uint32_t> counts(n);
    HIP_TRY(hipMemcpyAsync(counts.data(), cnt.p, (size_t)n * 4, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (out_npeaks) std::copy(counts.begin(), counts.end(), out_npeaks);
    std::vector<uint64_t> off((size_t)n + 1, 0);
    uint32_t pcap = 1;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t c = counts[i] >= min_peaks ? counts[i] : 0;  // runner.rs:313: too few peaks -> not searched
        off[i + 1] = off[i] + c;
        pcap = std::max(pcap, c);
    }
    HIP_TRY(d->peak_off.upload(off.data(), n ? (size_t)n + 1 : 0));
    HIP_TRY(d->masses.alloc(off[n]));
    HIP_TRY(d->intensities.alloc(off[n]));
    launch_compact(n, d->peak_off.p, stride, sm.p, si.p, d->masses.p, d->intensities.p, s->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s->stream));
    int rc = finish_batch(s, d.get(), n, raw->precursor_mz, raw->precursor_charge, raw->isolation_lo, raw->isolation_hi,
                          raw->scan_start_time, raw->inverse_ion_mobility, raw->file_id, pcap);
    if (rc != SAGE_HIP_OK) return rc;
    *out = d.release();
    return SAGE_HIP_OK;
}

int sage_hip_batch_download(SageDeviceBatch* b, uint64_t* peak_off, float* masses, float* intensities, float* tic) {
    if (!b || !peak_off) return fail(SAGE_HIP_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(b->device));
    if (b->n) HIP_TRY(hipMemcpy(peak_off, b->peak_off.p, ((size_t)b->n + 1) * 8, hipMemcpyDeviceToHost));
    else peak_off[0] = 0;
    const uint64_t total = b->n ? peak_off[b->n] : 0;
    if (masses && total) HIP_TRY(hipMemcpy(masses, b->masses.p, total * 4, hipMemcpyDeviceToHost));
    if (intensities && total) HIP_TRY(hipMemcpy(intensities, b->intensities.p, total * 4, hipMemcpyDeviceToHost));
    if (tic && b->n) HIP_TRY(hipMemcpy(tic, b->tic.p, (size_t)b->n * 4, hipMemcpyDeviceToHost));
    return SAGE_HIP_OK;
}

void sage_hip_batch_free(SageDeviceBatch* b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    delete b;
}

static int ensure_work(SageScorer* s, uint32_t n) {
    if (n <= s->work_n) return SAGE_HIP_OK;
    HIP_TRY(s->cand.alloc((size_t)n * s->dev.kmax));
    HIP_TRY(s->cand_len.alloc(n));
    HIP_TRY(s->totals.alloc((size_t)n * 2));
    HIP_TRY(s->status.alloc(n));
    HIP_TRY(s->queue.alloc(n));
    HIP_TRY(s->retry.alloc(n));
    HIP_TRY(s->out_count.alloc(n));
    // large-window pipeline: per-query records, verbatim slots, replayed heaps, and the candidate arena
    // (16 Ki entries = 64 KiB per spectrum on average; an exhausted arena is reported, never silently truncated)
    HIP_TRY(s->qrec.alloc((size_t)n * s->qmax));
    HIP_TRY(s->seeds.alloc((size_t)n * s->qmax * 64));
    HIP_TRY(s->qres.alloc((size_t)n * s->qmax * 64));
    // + two 64 Ki-entry chunks per resident workgroup (each takes its arena space a chunk at a time)
    uint64_t arena_entries = std::max<uint64_t>((uint64_t)n * 16384, 16ull << 20) + (uint64_t)std::min(n, s->tile_blocks) * (2u << 16);
    arena_entries = std::min<uint64_t>(arena_entries, 0xFFFFFFF0ull);
    if (const char* e = getenv("SAGE_HIP_ARENA_MB")) arena_entries = std::min<uint64_t>((uint64_t)std::max(1, atoi(e)) << 18, 0xFFFFFFF0ull);
    if (arena_entries > s->arena.n) HIP_TRY(s->arena.alloc(arena_entries));
    HIP_TRY(s->features.alloc((size_t)n * s->params.report_psms));
    s->work_n = n;
    return SAGE_HIP_OK;
}

static DevWork make_work(SageScorer* s);

// One pass of the kernels over `view` (the whole resident batch, or — `exact` retry pass — the spectra listed in view.order).
// exact: every trim replays bounded_min_heapify (reference heap layouts); otherwise the order-free trims.
static int run_kernels(SageScorer* s, SageDeviceBatch* b, bool with_rescore, bool exact, const DevBatchView* sub = nullptr) {
    if (b->device != s->db->device) return fail(SAGE_HIP_ERR_INVALID, "batch and scorer live on different devices");
    HIP_TRY(hipSetDevice(s->db->device));
    int rc = ensure_work(s, b->n);
    if (rc != SAGE_HIP_OK) return rc;
    const DevBatchView& view = sub ? *sub : b->view;
    DevScorer sc = s->dev;
    sc.exact = exact ? 1u : 0u;
    const size_t lds_p = prelim_lds_bytes(sc, view), lds_r = rescore_lds_bytes(sc, view, s->db->max_ions, true);
    const size_t lds_t = tile_lds_bytes(s->db->view, sc, view);
    if (lds_p > 64 * 1024 || lds_r > 64 * 1024 || lds_t > 160 * 1024)
        return fail(SAGE_HIP_ERR_UNSUPPORTED, "spectrum too large for the LDS staging of this build (peaks x fragment charges)");
    DevWork w = make_work(s);
    {
        TileParams tp{s->db->view, sc, view, w};
        HIP_TRY(hipMemcpyAsync(s->tile_params.p, &tp, sizeof tp, hipMemcpyHostToDevice, s->stream));  // (small: staged at call time)
    }
    HIP_TRY(hipMemsetAsync(s->n_deferred.p, 0, CTR_COUNT * 4, s->stream));
    HIP_TRY(hipEventRecord(s->ev[0], s->stream));
    launch_prelim(s->db->view, sc, view, w, s->stream);
    HIP_TRY(hipGetLastError());  // (a failed launch must not let the kernels downstream of it run on stale records)
    launch_prelim_tile(s->db->view, sc, view, w, s->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s->ev[1], s->stream));
    if (with_rescore)
        launch_rescore(s->db->view, sc, view, w, s->lnfact.p, (uint32_t)s->lnfact.n, s->db->max_ions,
                       s->features.p, s->out_count.p, nullptr, s->stream);
    HIP_TRY(hipEventRecord(s->ev[2], s->stream));
    HIP_TRY(hipMemcpyAsync(s->h_counters, s->n_deferred.p, CTR_COUNT * 4, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipGetLastError());
    return SAGE_HIP_OK;
}

static int finish_timing(SageScorer* s, bool with_rescore) {
    HIP_TRY(hipEventSynchronize(s->ev[2]));
    float a = 0, c = 0;
    HIP_TRY(hipEventElapsedTime(&a, s->ev[0], s->ev[1]));
    HIP_TRY(hipEventElapsedTime(&c, s->ev[1], s->ev[2]));
    s->timing.prelim_ms = a;
    s->timing.rescore_ms = with_rescore ? c : 0.f;
    s->timing.total_ms = a + c;
    s->timing.n_launches = with_rescore ? 6 : 5;
    s->timing.n_wide = s->h_counters[CTR_QUEUED];  // copied on the stream before the caller's synchronize
    s->timing.arena_entries = s->h_counters[CTR_ARENA_PTR];
    s->timing.n_retry = 0;
    if (s->h_counters[CTR_ARENA_OVERFLOW])
        return fail(SAGE_HIP_ERR_UNSUPPORTED, "large-window candidate arena exhausted (" + std::to_string(s->arena.n >> 18) +
                                                  " MiB): score this batch in smaller pieces or raise SAGE_HIP_ARENA_MB");
    return SAGE_HIP_OK;
}

static DevWork make_work(SageScorer* s) {
    DevWork w{};
    w.cand = s->cand.p;
    w.cand_len = s->cand_len.p;
    w.totals = s->totals.p;
    w.status = s->status.p;
    w.n_deferred = s->n_deferred.p;
    w.queue = s->queue.p;
    w.retry = s->retry.p;
    w.tile_blocks = s->tile_blocks;
    w.qrec = s->qrec.p;
    w.seeds = s->seeds.p;
    w.qres = s->qres.p;
    w.arena = s->arena.p;
    w.arena_cap = (uint32_t)s->arena.n;
    w.qmax = s->qmax;
    w.dbg = s->dbg.p;
    w.tile_params = s->tile_params.p;
    return w;
}

int sage_hip_score_resident(SageScorer* s, SageDeviceBatch* b, SageFeature* out, uint32_t* out_count) {
    if (!s || !b || !out || !out_count) return fail(SAGE_HIP_ERR_INVALID, "null argument");
    const bool exact = s->exact_always;
    int rc = run_kernels(s, b, true, exact);
    if (rc != SAGE_HIP_OK) return rc;
    // the counters (copied on the stream by run_kernels) decide whether a second pass is needed before anything is downloaded
    HIP_TRY(hipStreamSynchronize(s->stream));
    rc = finish_timing(s, true);
    if (rc != SAGE_HIP_OK) return rc;
    const uint32_t n_retry = exact ? 0 : s->h_counters[CTR_RETRY];
    if (n_retry && !s->h_counters[CTR_LIST_OVERFLOW]) {
        // Some spectra have equal hyperscores at a reported rank: there the heap layout of the preliminary list decides
        // the order (the stable sort of scoring.rs:495), so exactly those spectra go through the kernels again with every
        // trim replaying bounded_min_heapify.
        const SageTiming first = s->timing;
        DevBatchView sub = b->view;
        sub.order = s->retry.p;  // filled by the rescoring kernel, in no particular order
        sub.n = n_retry;
        rc = run_kernels(s, b, true, true, &sub);
        if (rc != SAGE_HIP_OK) return rc;
        HIP_TRY(hipStreamSynchronize(s->stream));
        rc = finish_timing(s, true);
        if (rc != SAGE_HIP_OK) return rc;
        s->timing.prelim_ms += first.prelim_ms;
        s->timing.rescore_ms += first.rescore_ms;
        s->timing.total_ms += first.total_ms;
        s->timing.n_launches += first.n_launches;
        s->timing.n_wide = first.n_wide;
        s->timing.arena_entries = std::max(s->timing.arena_entries, first.arena_entries);
        s->timing.n_retry = n_retry;
    }
    HIP_TRY(hipMemcpyAsync(out_count, s->out_count.p, (size_t)b->n * 4, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipMemcpyAsync(out, s->features.p, (size_t)b->n * s->params.report_psms * sizeof(SageFeature),
                           hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->h_counters[CTR_LIST_OVERFLOW]) {  // rare: find the offending spectrum for the message
        std::vector<uint32_t> st(b->n);
        HIP_TRY(hipMemcpy(st.data(), s->status.p, (size_t)b->n * 4, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < b->n; i++)
            if (st[i] != ST_OK)
                return fail(SAGE_HIP_ERR_UNSUPPORTED, "spectrum " + std::to_string(i) + ": preliminary pass status " +
                                                          std::to_string(st[i]) + " (candidate list capacity)");
    }
    return SAGE_HIP_OK;
}

int sage_hip_score_batch(SageScorer* s, const SageSpectrumBatch* batch, SageFeature* out, uint32_t* out_count) {
    SageDeviceBatch* d = nullptr;
    int rc = sage_hip_batch_upload(s, batch, &d);
    if (rc != SAGE_HIP_OK) return rc;
    rc = sage_hip_score_resident(s, d, out, out_count);
    sage_hip_batch_free(d);
    return rc;
}

int sage_hip_annotate_resident(SageScorer* s, SageDeviceBatch* b, const SageFeature* features, const uint32_t* counts,
                               SageFragments* out) {
    if (!s || !b || !features || !counts || !out || !out->psm_off) return fail(SAGE_HIP_ERR_INVALID, "null argument");
    if (b->device != s->db->device) return fail(SAGE_HIP_ERR_INVALID, "batch and scorer live on different devices");
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
    DevBuf<SageFeature> d_feats;
    DevBuf<uint32_t> d_counts;
    DevBuf<uint64_t> d_off;
    DevBuf<uint8_t> d_kinds;
    DevBuf<int32_t> d_charges, d_ord;
    DevBuf<float> d_int, d_calc, d_exp;
    HIP_TRY(d_feats.upload(features, slots));
    HIP_TRY(d_counts.upload(counts, n));
    HIP_TRY(d_off.upload(out->psm_off, slots + 1));
    HIP_TRY(d_kinds.alloc(total));
    HIP_TRY(d_charges.alloc(total));
    HIP_TRY(d_ord.alloc(total));
    HIP_TRY(d_int.alloc(total));
    HIP_TRY(d_calc.alloc(total));
    HIP_TRY(d_exp.alloc(total));
    DevFragments df{total, d_kinds.p, d_charges.p, d_ord.p, d_int.p, d_calc.p, d_exp.p};
    launch_annotate(s->db->view, s->dev, b->view, d_feats.p, d_counts.p, d_off.p, df, s->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipMemcpy(out->kinds, d_kinds.p, total, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out->charges, d_charges.p, total * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out->fragment_ordinals, d_ord.p, total * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out->intensities, d_int.p, total * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out->mz_calculated, d_calc.p, total * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out->mz_experimental, d_exp.p, total * 4, hipMemcpyDeviceToHost));
    return SAGE_HIP_OK;
}

int sage_hip_quick_score_resident(SageScorer* s, SageDeviceBatch* b, int prefilter_low_memory, uint8_t* keep) {
    if (!s || !b || !keep) return fail(SAGE_HIP_ERR_INVALID, "null argument");
    int rc = run_kernels(s, b, false, false);  // Scorer::initial_hits; which peptides survive each trim does not depend on heap layouts
    if (rc != SAGE_HIP_OK) return rc;
    const uint64_t np = s->db->view.np;
    DevBuf<uint8_t> d_keep;
    HIP_TRY(d_keep.alloc(np));
    HIP_TRY(hipMemsetAsync(d_keep.p, 0, std::max<uint64_t>(np, 1), s->stream));
    const DevWork w = make_work(s);
    if (prefilter_low_memory)
        launch_rescore(s->db->view, s->dev, b->view, w, s->lnfact.p, (uint32_t)s->lnfact.n, s->db->max_ions, s->features.p,
                       s->out_count.p, d_keep.p, s->stream);
    else
        launch_quick_mark(s->dev, b->view, w, d_keep.p, s->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s->stream));
    rc = finish_timing(s, false);
    if (rc != SAGE_HIP_OK) return rc;
    if (s->h_counters[CTR_LIST_OVERFLOW]) return fail(SAGE_HIP_ERR_UNSUPPORTED, "preliminary candidate list capacity exceeded");
    std::vector<uint8_t> h(np);
    HIP_TRY(hipMemcpy(h.data(), d_keep.p, np, hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < np; i++) keep[i] |= h[i];
    return SAGE_HIP_OK;
}

int sage_hip_initial_hits(SageScorer* s, SageDeviceBatch* b, uint64_t* packed, uint32_t cap, uint32_t* len,
                          uint64_t* matched_peaks, uint64_t* scored_candidates) {
    if (!s || !b || !packed || !len) return fail(SAGE_HIP_ERR_INVALID, "null argument");
    if (cap < s->dev.kmax) return fail(SAGE_HIP_ERR_INVALID, "cap must be >= max(50, 2*report_psms)");
    int rc = run_kernels(s, b, false, true);  // the reference's heap layouts
    if (rc != SAGE_HIP_OK) return rc;
    HIP_TRY(hipStreamSynchronize(s->stream));
    rc = finish_timing(s, false);
    if (rc != SAGE_HIP_OK) return rc;
    const uint32_t n = b->n, kmax = s->dev.kmax;
    std::vector<uint64_t> c((size_t)n * kmax);
    std::vector<uint32_t> tot((size_t)n * 2), st(n);
    HIP_TRY(hipMemcpy(c.data(), s->cand.p, c.size() * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(len, s->cand_len.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(tot.data(), s->totals.p, tot.size() * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(st.data(), s->status.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; i++) {
        if (st[i] != ST_OK) return fail(SAGE_HIP_ERR_UNSUPPORTED, "spectrum " + std::to_string(i) + ": status " + std::to_string(st[i]));
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
    return SAGE_HIP_OK;
}
void sage_hip_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int sage_hip_last_timing(const SageScorer* s, SageTiming* out) {
    if (!s || !out) return fail(SAGE_HIP_ERR_INVALID, "null argument");
    *out = s->timing;
    return SAGE_HIP_OK;
}

}  // extern "C"
