// Plain-data views passed by value to the HIP kernels (kernels.hip) by the C-ABI layer (capi.hip).
#pragma once
#include <stdint.h>

#include "../../include/sage_hip.h"
#include "core.h"

namespace sagehip {

// Device-resident database.
struct DevDbView {
    const float* pep_mono;        // [np]   peptide masses, ascending — the precursor-window search key
    uint32_t np;
    // peptide-major copy of IndexedDatabase.fragments: the same (peptide_index, fragment_mz) entries,
    // grouped by peptide so that a precursor window is ONE contiguous, coalesced range
    const SageTheoretical* pm_frag;  // [nf]
    const uint64_t* pm_off;          // [np + 1]
    // complete ion table for rescoring: ions[ion_off[p] + k*(L-1) + idx] = IonSeries(p, kinds[k])[idx]
    const float* ions;
    const uint64_t* ion_off;         // [np + 1]
    const uint32_t* pep_info;        // [np] len | decoy<<16 | missed_cleavages<<24
    // m/z-major copy of the fragments (globally ascending m/z) + a position table for open / wide-window
    // searches: mz_lut[b] = #fragments with m/z < b / lut_scale
    const SageTheoretical* mz_frag;  // [nf]
    const uint32_t* mz_lut;          // [lut_n]
    uint32_t lut_n;
    float lut_scale;
    uint64_t nf;
    uint8_t ion_kinds[8];
    uint32_t n_kinds;
};

struct DevScorer {
    sagecore::Tol precursor_tol, fragment_tol;
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
    uint32_t list_cap;   // capacity (entries) of each of the two CLists
    uint32_t wcap;       // candidate-slot capacity of the LDS counter array (narrow path)
    uint32_t dbg_flags;    // timing experiments only (SAGE_HIP_DEBUG_FLAGS): results are WRONG when non-zero
    uint32_t open_thresh;  // windows with more candidate slots than this use the m/z-major (open-search) kernel
};

struct DevBatchView {
    uint32_t n;
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
    uint32_t pcap;              // max peaks per spectrum in this batch
    uint32_t fzcap;             // max (max_fragment_charge - 1) over the charges this batch can use
};

struct DevWork {  // per-spectrum outputs of the preliminary pass
    uint64_t* cand;        // [n * kmax] packed PreScore in the reference's heap-layout order
    uint32_t* cand_len;    // [n]
    uint32_t* totals;      // [n * 2] InitialHits.matched_peaks, .scored_candidates
    uint32_t* status;      // [n] 0 ok, 1 deferred to the large-window path, 2 list overflow
    uint32_t* n_deferred;  // [4]: [0] spectra deferred to the mid-window kernel, [1] list overflows,
                           //      [2] spectra deferred to the open-search kernel
    uint32_t* wide_cnt;    // [wide_blocks * (np + 1)] global counter scratch for the large-window path
    uint32_t wide_blocks;
    uint64_t wide_words;   // counter slots per mid-window block
    uint32_t* open_cnt;    // [open_blocks * open_words] u16-pair counters, all-zero between spectra
    uint32_t open_blocks;
    uint64_t open_words;
    unsigned long long* dbg;  // optional [2][8] per-phase cycle accumulators (null in production)
};

enum { ST_OK = 0, ST_DEFERRED = 1, ST_OVERFLOW = 2, ST_DEFERRED_OPEN = 3 };

// launch wrappers (kernels.hip)
size_t prelim_lds_bytes(const DevScorer& sc, const DevBatchView& b, bool wide);
uint32_t rescore_item_cap(const DevBatchView& b, uint32_t max_ions);
size_t rescore_lds_bytes(const DevScorer& sc, const DevBatchView& b, uint32_t max_ions);
void launch_prelim(const DevDbView& db, const DevScorer& sc, const DevBatchView& b, const DevWork& w, void* stream);
void launch_prelim_wide(const DevDbView& db, const DevScorer& sc, const DevBatchView& b, const DevWork& w, void* stream);
void launch_prelim_open(const DevDbView& db, const DevScorer& sc, const DevBatchView& b, const DevWork& w, void* stream);
void launch_rescore(const DevDbView& db, const DevScorer& sc, const DevBatchView& b, const DevWork& w,
                    const double* lnfact_table, uint32_t lnfact_n, uint32_t max_ions, SageFeature* out,
                    uint32_t* out_count, void* stream);

}  // namespace sagehip
