/*
 * sage_hip.h — C ABI of the MI355X-native search-and-score engine (libsage_hip.so).
 *
 * This is the drop-in boundary for ONE path of lazear/sage: `Scorer::score` /
 * `score_chimera_fast` over `IndexedDatabase` (reference citations are relative to
 * /root/reference/crates/sage/src unless a crate is named).  The reference has no FFI; the entry
 * points below are what a Rust `extern "C"` shim inside sage-core would bind (INTEGRATION.md shows
 * the shim).  Conventions:
 *   - plain pointers and sizes only; the caller owns every input buffer for the duration of a call;
 *     outputs are caller-allocated;
 *   - every function returns a status code (SAGE_HIP_OK == 0) instead of panicking
 *     (the reference panics at scoring.rs:261-267, 301-304, 466-468); sage_hip_last_error() gives
 *     the message of the last failure on the calling thread;
 *   - a SageScorer handle may be shared by any number of host threads, like `&Scorer` (scoring.rs:300 is called from every
 *     rayon worker): calls on ONE handle run one after the other; sage_hip_scorer_clone() gives a thread its own handle
 *     (own streams and working set, same device database) when batches should be scored concurrently.  Use one device
 *     database + scorer per device for multi-GPU (spectra sharded, index replicated, no collective on the data path).
 *   - there is NO CPU fallback: scoring entry points fail with SAGE_HIP_ERR_NO_DEVICE when no
 *     gfx950 device is usable.
 */
#ifndef SAGE_HIP_H
#define SAGE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAGE_HIP_ABI_VERSION 6

enum {
    SAGE_HIP_OK = 0,
    SAGE_HIP_ERR_INVALID = 1,     /* bad argument / contract violation (reference: panic!) */
    SAGE_HIP_ERR_NO_DEVICE = 2,   /* no usable HIP device */
    SAGE_HIP_ERR_HIP = 3,         /* a HIP runtime call failed */
    SAGE_HIP_ERR_UNSUPPORTED = 4, /* valid in the reference, outside this build's device limits */
    SAGE_HIP_ERR_OOM = 5,
    SAGE_HIP_ERR_INTERNAL = 6     /* an invariant of the library itself did not hold (a bug: report it) */
};

/* mass.rs:10-16  enum Tolerance { Ppm(f32,f32), Pct(f32,f32), Da(f32,f32) } */
enum { SAGE_TOL_PPM = 0, SAGE_TOL_PCT = 1, SAGE_TOL_DA = 2 };
typedef struct SageTolerance {
    int32_t kind;
    float lo, hi;
} SageTolerance;

/* ion_series.rs:6-15  enum Kind { A, B, C, X, Y, Z } */
enum { SAGE_ION_A = 0, SAGE_ION_B = 1, SAGE_ION_C = 2, SAGE_ION_X = 3, SAGE_ION_Y = 4, SAGE_ION_Z = 5 };

/* database.rs:378-382  struct Theoretical */
typedef struct SageTheoretical {
    uint32_t peptide_index; /* PeptideIx, database.rs:367-369 */
    float fragment_mz;
} SageTheoretical;

/* ------------------------------------------------------------------------------------------
 * Host side: database construction.  Replaces database.rs:59-139 (Builder / Parameters) and
 * Parameters::build (database.rs:260-364).  Option<T> fields use -1 / NULL for None.
 * ---------------------------------------------------------------------------------------- */
typedef struct SageDbParams {
    uint64_t bucket_size;      /* 0 => 8192; rounded up to a power of two (database.rs:97) */
    int32_t missed_cleavages;  /* EnzymeBuilder fields, database.rs:15-27 */
    int32_t min_len, max_len;
    const char* cleave_at;
    const char* restrict_;
    int32_t c_terminal;
    int32_t semi_enzymatic;
    int32_t enzyme_present;    /* 0 => the whole `enzyme` object is absent (database.rs:105) */
    float peptide_min_mass, peptide_max_mass;
    const uint8_t* ion_kinds;  /* SAGE_ION_* ; NULL/0 => [b, y] */
    uint32_t n_ion_kinds;
    uint64_t min_ion_index;
    const char* const* static_mod_keys; /* "C", "^", "$K", "[", ... (modification.rs:66-104) */
    const float* static_mod_masses;
    uint32_t n_static_mods;
    const char* const* var_mod_keys;    /* one entry per (key, mass) pair */
    const float* var_mod_masses;
    uint32_t n_var_mods;
    uint64_t max_variable_mods;
    const char* decoy_tag;
    int32_t generate_decoys;
    int32_t peptides_only;     /* 1 => stop after reorder_peptides (database.rs:221-258): no fragments / min_value; the index is
                                  then generated on the device by sage_hip_db_create (build_from_peptides, :265-346) */
} SageDbParams;

/* Flat, read-only view of an IndexedDatabase (database.rs:384-395) + the Peptide fields the path
 * reads (peptide.rs:12-31).  nterm/cterm: NaN == None. */
typedef struct SageDbView {
    const SageTheoretical* fragments; /* globally m/z sorted, then peptide-sorted inside each bucket */
    uint64_t n_fragments;
    const float* min_value;           /* [n_buckets] */
    uint64_t n_buckets;
    uint64_t bucket_size;
    const float* pep_mono;            /* [n_peptides], ascending (total_cmp) */
    const uint64_t* seq_off;          /* [n_peptides + 1] */
    const uint8_t* seq;               /* residues, ASCII */
    const float* mods;                /* per-residue modification mass, parallel to seq */
    const float* nterm;
    const float* cterm;
    const uint8_t* decoy;
    const uint8_t* missed_cleavages;
    uint64_t n_peptides;
    const uint8_t* ion_kinds;
    uint32_t n_ion_kinds;
    uint64_t min_ion_index;           /* read only when fragments == NULL (device-side build_from_peptides) */
} SageDbView;

typedef struct SageHostDb SageHostDb;

/* Parameters::build(Fasta::parse(text)) — database.rs:260-263, fasta.rs:16-56 */
int sage_hip_hostdb_build(const char* fasta_text, const SageDbParams* params, SageHostDb** out);
/* ---- the `prefilter` flow of sage-cli (runner.rs:104-127, :143-238) ----------------------------------------------------
 * database.prefilter: the FASTA is searched in chunks of `prefilter_chunk_size` target proteins with Scorer::quick_score
 * (sage_hip_quick_score_resident), and the final database holds only the peptides some spectrum picked. */
/* Fasta::parse(..).targets.len() (fasta.rs:16-56) */
int sage_hip_fasta_num_targets(const char* fasta_text, const SageDbParams* params, uint64_t* out);
/* Parameters::auto_calculate_prefilter_chunk_size (database.rs:142-160); `requested` = the configured value, 0 = auto */
int sage_hip_prefilter_chunk_size(const char* fasta_text, const SageDbParams* params, uint64_t requested, uint64_t* out);
/* Parameters::build over targets [first_target, first_target + n_targets): one chunk of Fasta::iter_chunks (fasta.rs:81-89) */
int sage_hip_hostdb_build_chunk(const char* fasta_text, const SageDbParams* params, uint64_t first_target, uint64_t n_targets,
                                SageHostDb** out);
/* runner.rs:215-238: the kept peptides of every chunk (keep[c][ix] != 0, consecutive chunks in FASTA order), through
 * Parameters::reorder_peptides and build_from_peptides (honours params->peptides_only) */
int sage_hip_hostdb_merge_kept(const SageHostDb* const* chunks, const uint8_t* const* keep, uint32_t n_chunks,
                               const SageDbParams* params, SageHostDb** out);
void sage_hip_hostdb_free(SageHostDb* db);
int sage_hip_hostdb_view(const SageHostDb* db, SageDbView* out);
/* Display string of peptide i ("[+42]-MEWK...", peptide.rs:391-408) and its ';'-joined proteins;
 * return the required buffer size including NUL. */
uint64_t sage_hip_hostdb_peptide_string(const SageHostDb* db, uint64_t i, char* out, uint64_t cap);
uint64_t sage_hip_hostdb_peptide_proteins(const SageHostDb* db, uint64_t i, char* out, uint64_t cap);
/* Peptide.proteins.len() and Peptide.semi_enzymatic (peptide.rs:28-30) — columns of results.sage.tsv */
int sage_hip_hostdb_peptide_info(const SageHostDb* db, uint64_t i, uint32_t* num_proteins, uint8_t* semi_enzymatic);

/* SpectrumProcessor::new(take_top_n, deisotope, min_deisotope_mz).process() for one centroided
 * MS2 spectrum (spectrum.rs:279-412).  precursor_charge 0 == None.  out_* need capacity n.
 * Returns the number of peaks kept. */
uint64_t sage_hip_process_ms2(uint64_t take_top_n, int deisotope, float min_deisotope_mz, const float* mz,
                              const float* intensity, uint64_t n, uint8_t precursor_charge, float* out_mass,
                              float* out_intensity, float* out_tic);

/* ------------------------------------------------------------------------------------------
 * Device side.
 * ---------------------------------------------------------------------------------------- */
typedef struct SageDeviceDb SageDeviceDb;
typedef struct SageScorer SageScorer;
typedef struct SageDeviceBatch SageDeviceBatch;

int sage_hip_device_count(void);

/* Upload an IndexedDatabase to HBM on `device` and derive the device layouts.
 * Stands in for the `&'db IndexedDatabase` borrow of Scorer (scoring.rs:211).
 * With view->fragments == NULL the fragment index is generated ON THE DEVICE from the peptide list
 * (Parameters::build_from_peptides, database.rs:265-346: ion series, stored-ion filter by view->min_ion_index, sort).
 * Peptides of any length up to 65 535 residues (beyond 1023 the general, slower rescoring instance scores the database: DESIGN.md 4.8;
 * the reference's default max_len is 50). */
int sage_hip_db_create(const SageDbView* view, int device, SageDeviceDb** out);
void sage_hip_db_destroy(SageDeviceDb* db);
uint64_t sage_hip_db_device_bytes(const SageDeviceDb* db);

/* scoring.rs:210-232  struct Scorer — every field, same meaning. */
typedef struct SageScorerParams {
    SageTolerance precursor_tol, fragment_tol;
    uint16_t min_matched_peaks;
    int8_t min_isotope_err, max_isotope_err;
    uint8_t min_precursor_charge, max_precursor_charge;
    uint8_t override_precursor_charge;
    uint8_t chimera;
    int16_t max_fragment_charge; /* Option<u8>: -1 == None */
    uint8_t wide_window;
    uint8_t annotate_matches;    /* the Fragments themselves are fetched with sage_hip_annotate_resident */
    uint32_t report_psms;        /* 1..32767 (above 32: the wider, slower kernels of DESIGN.md 4.8; lists that do not fit a compute unit's
                                  * LDS live in a global-memory workspace) */
    int32_t score_type;          /* 0 SageHyperScore, 1 OpenMSHyperScore (scoring.rs:10-14) */
} SageScorerParams;

int sage_hip_scorer_create(SageDeviceDb* db, const SageScorerParams* params, SageScorer** out);
/* A second handle with the same parameters on the same device database (`&Scorer` is Sync: runner.rs:311-325 calls score from
 * every worker thread).  Handles are independent: own streams, own device working set. */
int sage_hip_scorer_clone(SageScorer* scorer, SageScorer** out);
void sage_hip_scorer_destroy(SageScorer* scorer);

/* A batch of ProcessedSpectrum (spectrum.rs:57-79) + precursors[0] (spectrum.rs:46-55), SoA.
 * All spectra must be MS2 (level == 2 is asserted by the reference at scoring.rs:301-304). */
typedef struct SageSpectrumBatch {
    uint32_t n_spectra;
    const uint64_t* peak_off;          /* [n + 1] */
    const float* masses;               /* ascending inside each spectrum */
    const float* intensities;
    const float* precursor_mz;         /* [n] precursors[0].mz */
    const uint8_t* precursor_charge;   /* [n] 0 == None */
    const float* isolation_lo;         /* [n] isolation_window = Tolerance::Da(lo, hi); NaN == None; may be NULL */
    const float* isolation_hi;
    const float* total_ion_current;    /* [n] */
    const float* scan_start_time;      /* [n] may be NULL (0) */
    const float* inverse_ion_mobility; /* [n] NaN == None; may be NULL */
    const uint32_t* file_id;           /* [n] may be NULL (0) */
} SageSpectrumBatch;

/* scoring.rs:69-149  struct Feature — the fields the hot path computes (the rest are defaults the
 * reference fills in later, scoring.rs:576-592).  psm_id (a global atomic, :163-167) is excluded. */
typedef struct SageFeature {
    uint32_t spec_index; /* position in the batch; stands in for spec_id */
    uint32_t peptide_idx;
    uint32_t rank;
    int32_t label;
    float expmass, calcmass, rt, ims, delta_mass, isotope_error, average_ppm;
    float longest_y_pct, matched_intensity_pct, ms2_intensity;
    double hyperscore, delta_next, delta_best, poisson;
    uint32_t matched_peaks, longest_b, longest_y, scored_candidates;
    uint32_t peptide_len, file_id;
    uint8_t charge, missed_cleavages;
    uint8_t pad[6];
} SageFeature;

/* Scorer::score for every spectrum of the batch (scoring.rs:300-309), results in input order:
 * out[i*report_psms + r] for r < out_count[i], Feature.spec_index == i.
 * Host memory in, host memory out, as a pipeline over chunks of the batch (SAGE_HIP_CHUNK spectra, default 131072): the
 * uploads run up to three chunks ahead on a copy stream while earlier chunks are scored (two at a time when the batch has no
 * large precursor windows) and their PSM records come back — the reader / processor / search overlap of runner.rs:365-375,
 * 450-461 at the PCIe boundary.  Arrays allocated with
 * sage_hip_host_alloc (page-locked) move by DMA at full PCIe rate; pageable arrays are accepted and staged through
 * page-locked blocks by a few host threads.  A chunk whose large-window candidates exhaust the device arena is scored again
 * in halves (never an error unless a single spectrum does not fit). */
int sage_hip_score_batch(SageScorer* scorer, const SageSpectrumBatch* batch, SageFeature* out,
                         uint32_t* out_count);

/* The same split in two so a batch can stay resident in HBM across calls.  `out` allocated with sage_hip_host_alloc
 * (page-locked, mapped) is written by the rescoring kernels themselves — no download phase; it holds the results when the
 * call returns.  Any other host memory gets a device buffer and a copy. */
int sage_hip_batch_upload(SageScorer* scorer, const SageSpectrumBatch* batch, SageDeviceBatch** out);
void sage_hip_batch_free(SageDeviceBatch* batch);
int sage_hip_score_resident(SageScorer* scorer, SageDeviceBatch* batch, SageFeature* out, uint32_t* out_count);

/* Raw centroided MS2 spectra (spectrum.rs:81-106 RawSpectrum + precursors[0]), SoA — the input of
 * SpectrumProcessor::process.  Same conventions as SageSpectrumBatch; mz ascending inside each spectrum. */
typedef struct SageRawBatch {
    uint32_t n_spectra;
    const uint64_t* peak_off;          /* [n + 1] */
    const float* mz;
    const float* intensities;
    const float* precursor_mz;         /* [n] */
    const uint8_t* precursor_charge;   /* [n] 0 == None (then fragments are deisotoped up to z = 3, spectrum.rs:289-293) */
    const float* isolation_lo;         /* [n] may be NULL */
    const float* isolation_hi;
    const float* scan_start_time;      /* [n] may be NULL */
    const float* inverse_ion_mobility; /* [n] may be NULL */
    const uint32_t* file_id;           /* [n] may be NULL */
} SageRawBatch;

/* SpectrumProcessor::new(take_top_n, deisotope, min_deisotope_mz).process() (spectrum.rs:279-412) for every spectrum of
 * the batch ON THE DEVICE, leaving the ProcessedSpectrum arrays resident for sage_hip_score_resident: raw peaks in, PSMs
 * out, no host round trip in between.  Spectra that keep fewer than `min_peaks` peaks are not searched (sage-cli
 * runner.rs:313): they stay in the batch with zero peaks and yield no PSM.  out_npeaks (optional, [n]) receives the
 * number of peaks each spectrum kept before that filter. */
int sage_hip_batch_process_upload(SageScorer* scorer, const SageRawBatch* raw, uint64_t take_top_n, int deisotope,
                                  float min_deisotope_mz, uint32_t min_peaks, SageDeviceBatch** out, uint32_t* out_npeaks);
/* Copy a resident batch's ProcessedSpectrum arrays back (tests, writers).  peak_off: [n + 1]; masses / intensities need
 * peak_off[n] entries — call once with masses == NULL to get peak_off first. */
int sage_hip_batch_download(SageDeviceBatch* batch, uint64_t* peak_off, float* masses, float* intensities, float* tic);

/* Scorer::initial_hits (scoring.rs:418-462) of every spectrum of a resident batch: the trimmed
 * preliminary list in the reference's heap-layout order.  packed[i*cap + j] =
 * matched<<48 | peptide<<16 | precursor_charge<<8 | (isotope_error+128); len[i] entries are valid. */
int sage_hip_initial_hits(SageScorer* scorer, SageDeviceBatch* batch, uint64_t* packed, uint32_t cap,
                          uint32_t* len, uint64_t* matched_peaks, uint64_t* scored_candidates);

/* scoring.rs:152-161  struct Fragments of every reported PSM (Scorer.annotate_matches, scoring.rs:722-752), flattened:
 * PSM r of spectrum i is slot s = i * report_psms + r and owns entries [psm_off[s], psm_off[s + 1]) of each array, in the
 * reference's push order (ion kind, ion index, fragment charge).  A slot's length equals its Feature.matched_peaks, so the
 * caller can size the arrays from the features; `capacity` is the number of entries available in each array. */
typedef struct SageFragments {
    uint64_t capacity;
    uint64_t* psm_off;            /* [n_spectra * report_psms + 1] */
    uint8_t* kinds;               /* SAGE_ION_* */
    int32_t* charges;
    int32_t* fragment_ordinals;
    float* intensities;
    float* mz_calculated;
    float* mz_experimental;
} SageFragments;
/* Annotate the PSMs `features` / `counts` that sage_hip_score_resident returned for this resident batch with this scorer
 * (with chimera, the winner's peaks are removed between PSMs exactly as score_chimera_fast does).  Fails with
 * SAGE_HIP_ERR_INVALID when `capacity` is too small; out->psm_off is filled either way. */
int sage_hip_annotate_resident(SageScorer* scorer, SageDeviceBatch* batch, const SageFeature* features,
                               const uint32_t* counts, SageFragments* out);

/* Scorer::quick_score (scoring.rs:255-298) for every spectrum of a resident batch — the first pass of the `prefilter`
 * flow (sage-cli runner.rs:143-240).  keep: host array [n_peptides] standing in for the reference's &[AtomicBool];
 * identified peptides are OR-ed in (keep[i] = 1). */
int sage_hip_quick_score_resident(SageScorer* scorer, SageDeviceBatch* batch, int prefilter_low_memory, uint8_t* keep);

/* Timing of the last sage_hip_score_resident / sage_hip_score_batch call, from HIP events recorded
 * on the scorer's own stream. */
typedef struct SageTiming {
    float prelim_ms;   /* fragment-match + k-select kernel(s) */
    float rescore_ms;  /* rescoring + top-K + Feature kernel */
    float total_ms;    /* first launch -> last kernel done (excludes H2D/D2H) */
    uint32_t n_launches;
    uint32_t n_wide;   /* spectra routed to the tiled large-window kernels */
    uint32_t arena_entries; /* 4-byte entries of the large-window candidate arena this call used */
    uint32_t n_retry;  /* spectra with equal hyperscores at a reported rank, re-run with exact heap layouts (the retry pass) */
    float retry_ms;    /* of total_ms: that retry pass */
    uint32_t n_tied;   /* SAGE_HIP_FUSED=1 only: narrow spectra with such a tie, settled inside the fused first-pass kernel */
    uint32_t n_ways;   /* sage_hip_score_resident: parts of the batch scored next to each other on their own streams (then
                        * prelim_ms / rescore_ms are sums over parts that overlap in time; total_ms is the wall span) */
} SageTiming;
int sage_hip_last_timing(const SageScorer* scorer, SageTiming* out);
/* The per-kernel times of SageTiming (prelim_ms / rescore_ms / retry_ms / total_ms) come from HIP events recorded between the
 * kernels of a sage_hip_score_resident call: eight records and six elapsed-time queries, ~15 us of a call — nothing next to a
 * 500 000-spectrum step, 2 % of a 62 500-spectrum one.  `every` = 1 (the default): every call is timed; n > 1: every n-th call,
 * the calls in between report the kernel times of the last timed call and total_ms == 0; 0: never.  The counters (n_retry, n_wide,
 * ...) are always the call's own.  No counterpart in the reference (runner.rs:327-330 times the whole search with a wall clock). */
int sage_hip_scorer_set_timing_interval(SageScorer* scorer, uint32_t every);

/* Page-locked host memory for `out` / `out_count`: results then arrive by DMA at full PCIe rate instead of
 * through the runtime's staging copy.  Optional — any host pointer is accepted by the scoring calls. */
int sage_hip_host_alloc(uint64_t bytes, void** out);
void sage_hip_host_free(void* p);

/* Debug aid, only with SAGE_HIP_PHASE_CLOCKS=1 at scorer creation: cumulative shader cycles per kernel phase over the
 * first 4096 work items, out32[8*k + phase]: k = 0 narrow preliminary kernel, 1 rescoring kernel, 2 large-window count
 * kernel, 3 large-window replay kernel. */
int sage_hip_debug_phase_cycles(SageScorer* scorer, unsigned long long* out32);

/* ---- post-search rescoring --------------------------------------------------------------
 * The step that consumes the Feature records of ALL searched files (sage-cli runner.rs:536-541):
 *   spectrum_fdr  (runner.rs:281-292)  = ml::linear_discriminant::score_psms (linear_discriminant.rs:133-231: mass-error KDE,
 *                                        20-feature LDA, Gauss-Jordan solve, projection, KDE posterior error) or the heuristic
 *                                        fall-back, then the sort by discriminant and ml::qvalue::spectrum_q_value (qvalue.rs:8-36)
 *   fdr::picked_peptide / picked_protein (fdr.rs:123-187) = Competition::assign_q_value (fdr.rs:60-120)
 * Row-parallel work (feature rows, class sums, scatter matrices, projection, kernel-density sums, sorts, scans) runs on the
 * device; the 20x20 solve and the bandwidth scalars are host arithmetic.  Retention-time / ion-mobility models
 * (ml/retention_model.rs, mobility_model.rs) are not part of this call: their outputs enter through the optional arrays. */
typedef struct SageRescoreInput {
    uint64_t n;                    /* Features of the whole run */
    const SageFeature* features;   /* host, [n] */
    const float* aligned_rt;       /* [n] or NULL: features[i].rt (Feature default, scoring.rs:576-592) */
    const float* delta_rt_model;   /* [n] or NULL: 0.999 */
    const float* delta_ims_model;  /* [n] or NULL: 0.999 */
    SageTolerance precursor_tol;   /* Ppm or Da (Pct is unreachable in the reference, linear_discriminant.rs:142) */
    /* picked competitions: dense ids 0..n_keys-1 (every id used at least once) of the map key the reference builds —
     * peptide: the sequence string, reversed for generated decoys (fdr.rs:126-132); protein: the proteins vector, only for
     * peptides with exactly one protein (fdr.rs:158-160), 0xFFFFFFFF otherwise (protein_q stays 1.0).
     * sage_hip_hostdb_competition_keys() produces both. */
    const uint32_t* peptide_key;   /* [n] */
    uint32_t n_peptide_keys;
    const uint32_t* protein_key;   /* [n] */
    uint32_t n_protein_keys;
} SageRescoreInput;

typedef struct SageRescoreOutput {
    /* caller-allocated host arrays [n], INPUT order */
    float* discriminant_score;
    float* posterior_error;        /* log10 PEP; 1.0 (the Feature default) when the linear model could not be fitted */
    float* spectrum_q;
    float* peptide_q;
    float* protein_q;
    uint32_t* order;               /* [n] or NULL: order[j] = input index of the j-th best PSM — the order the reference
                                      leaves `features` in (runner.rs:290) and writes them out */
    /* filled by the call */
    uint64_t passing_spectrum;     /* PSMs with q <= 0.01 (targets and decoys, qvalue.rs:31) */
    uint64_t passing_peptide;      /* target peptides at 1 % (fdr.rs:105) */
    uint64_t passing_protein;
    int32_t lda_fitted;            /* 0: heuristic discriminant of runner.rs:285-288 */
    double coef[20];               /* LDA coefficients, FEATURE_NAMES order (linear_discriminant.rs:20-41) */
    float device_ms;               /* HIP-event time of all kernels of the call */
} SageRescoreOutput;

int sage_hip_rescore(int device, const SageRescoreInput* in, SageRescoreOutput* out);

/* ---- the predict_rt block of sage-cli (runner.rs:513-530), the producer of SageRescoreInput's optional arrays ----------
 *   features.par_sort_unstable_by(poisson) + spectrum_q_value  (runner.rs:517-520, qvalue.rs:8-36): the training filter
 *   ml::retention_alignment::global_alignment                  (retention_alignment.rs:100-173)  -> aligned_rt
 *   ml::retention_model::predict                               (retention_model.rs:14-90, regression.rs:58-122)
 *   ml::mobility_model::predict                                (mobility_model.rs:14-186)
 * Peptide data the two models embed comes per Feature: sequence bytes and monoisotopic mass of db[features[i].peptide_idx]
 * (sage_hip_hostdb_feature_peptides gathers them). */
typedef struct SageRtInput {
    uint64_t n;
    const SageFeature* features;   /* host, [n]: rt, ims, charge, label, file_id, poisson, peptide_idx are read */
    uint32_t n_files;
    const uint64_t* seq_off;       /* [n + 1] */
    const uint8_t* seq;            /* residues, 'A'..'Z' */
    const float* monoisotopic;     /* [n] */
} SageRtInput;

typedef struct SageAlignment {     /* retention_alignment.rs:92-98 */
    uint32_t file_id;
    float max_rt, slope, intercept;
} SageAlignment;

typedef struct SageRtOutput {
    /* caller-allocated host arrays [n], input order; Feature defaults (scoring.rs:576-592) where a model is not fitted */
    float* spectrum_q;             /* q-values of the poisson-sorted pass */
    float* aligned_rt;
    float* predicted_rt;
    float* delta_rt_model;
    float* predicted_ims;
    float* delta_ims_model;
    SageAlignment* alignments;     /* [n_files] */
    /* filled by the call */
    int32_t rt_fitted, ims_fitted; /* LinearRegression::fit returned Some */
    double rt_r2, ims_r2;
    float device_ms;
} SageRtOutput;

int sage_hip_predict_rt(int device, const SageRtInput* in, SageRtOutput* out);

/* Peptide.sequence / .monoisotopic of db[peptide_idx[i]] for SageRtInput.  Call with seq == NULL to size: seq_off is filled. */
int sage_hip_hostdb_feature_peptides(const SageHostDb* db, const uint32_t* peptide_idx, uint64_t n, uint64_t* seq_off,
                                     uint8_t* seq, float* monoisotopic);

/* ---- mzML input (host; sage-cloudpath/src/mzml.rs:109-403 MzMLReader::with_file_id_and_level_filter(..).parse): the MSn
 * spectra of one file as a SageRawBatch whose arrays the handle owns, ready for sage_hip_batch_process_upload.  ms_level < 0
 * keeps every level.  spectrum ids: sage_hip_mzml_spectrum_id. */
typedef struct SageMzml SageMzml;
int sage_hip_mzml_read(const char* path, uint32_t file_id, int ms_level, SageMzml** out);
/* The inputs the reference refuses to search — it panics while PROCESSING them, the reader accepts them: an MS2 spectrum in
 * profile mode (spectrum.rs:280-286; Representation defaults to Profile, so the centroid term MS:1000127 must be present) or
 * without a precursor (scoring.rs:466-468).  SAGE_HIP_ERR_INVALID with the reference's message; call before searching a run. */
int sage_hip_mzml_check_searchable(const SageMzml* run);
int sage_hip_mzml_view(const SageMzml* run, SageRawBatch* out);
const char* sage_hip_mzml_spectrum_id(const SageMzml* run, uint64_t i);
void sage_hip_mzml_free(SageMzml* run);

/* ---- writers (host): results.sage.tsv / results.sage.pin, byte for byte as sage-cli/src/runner.rs:687-780, :830-905,
 * :938-1135 format them (itoa integers, ryu floats).  Arrays of SagePostColumns may be NULL: the Feature defaults of
 * scoring.rs:576-592 are written (aligned_rt = rt, predicted_* 0.0, delta_*_model 0.999, discriminant 0.0, posterior_error
 * and q-values 1.0).  protein-group columns are always the defaults. */
typedef struct SagePostColumns {
    const float* discriminant_score;
    const float* posterior_error;
    const float* spectrum_q;
    const float* peptide_q;
    const float* protein_q;
    const float* aligned_rt;
    const float* predicted_rt;
    const float* delta_rt_model;
    const float* predicted_ims;
    const float* delta_ims_model;
} SagePostColumns;
enum { SAGE_FORMAT_TSV = 0, SAGE_FORMAT_PIN = 1 };
/* order: [n] row order (indices into features) or NULL; psm_id, spec_ids (spectrum ids, NUL-terminated): [n], indexed like
 * features; filenames: [n_files], indexed by SageFeature.file_id. */
int sage_hip_write_results(const char* path, int format, const SageHostDb* db, const SageFeature* features, uint64_t n,
                           const uint64_t* order, const uint64_t* psm_id, const char* const* filenames, uint32_t n_files,
                           const char* const* spec_ids, const SagePostColumns* post);

/* The competition keys of SageRescoreInput for `n` PSMs given their peptide indices (host work: string keys). */
int sage_hip_hostdb_competition_keys(const SageHostDb* db, const uint32_t* peptide_idx, uint64_t n, uint32_t* peptide_key,
                                 uint32_t* n_peptide_keys, uint32_t* protein_key, uint32_t* n_protein_keys);

const char* sage_hip_last_error(void);
int sage_hip_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SAGE_HIP_H */
