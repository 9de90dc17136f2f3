/* drive_search.c — the search-and-score path driven from plain C through include/sage_hip.h, no Python in the data path:
 *   FASTA text -> sage_hip_hostdb_build (peptides only) -> sage_hip_db_create (index generated on the device)
 *   mzML file  -> sage_hip_mzml_read -> sage_hip_batch_process_upload (SpectrumProcessor::process on the device)
 *   -> sage_hip_score_resident -> sage_hip_rescore (LDA, q-values, picked FDR) -> sage_hip_write_results (results.sage.tsv)
 * The parameters are the defaults of sage-cli's input.rs with trypsin / 1 missed cleavage / C+57.0215 / decoys.
 * tests/test_cli_io.py builds it with gcc (-lsage_hip), runs it and compares the file byte for byte with the Python CLI's.
 *
 *   usage: drive_search <fasta> <mzML> <out.tsv>
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sage_hip.h"

#define CHECK(call)                                                                        \
    do {                                                                                   \
        int rc_ = (call);                                                                  \
        if (rc_ != SAGE_HIP_OK) {                                                          \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, sage_hip_last_error());    \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

static char* slurp(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char* s = (char*)malloc((size_t)n + 1);
    if (s && fread(s, 1, (size_t)n, f) != (size_t)n) { free(s); s = NULL; }
    if (s) s[n] = 0;
    fclose(f);
    return s;
}

int main(int argc, char** argv) {
    if (argc != 4) {
        fprintf(stderr, "usage: %s <fasta> <mzML> <out.tsv>\n", argv[0]);
        return 2;
    }
    if (sage_hip_abi_version() != SAGE_HIP_ABI_VERSION) {
        fprintf(stderr, "header / library ABI mismatch\n");
        return 2;
    }
    char* fasta = slurp(argv[1]);
    if (!fasta) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }

    /* database.rs:59-139 Builder with an explicit enzyme object */
    const uint8_t kinds[2] = {SAGE_ION_B, SAGE_ION_Y};
    const char* static_keys[1] = {"C"};
    const float static_masses[1] = {57.0215f};
    SageDbParams dbp;
    memset(&dbp, 0, sizeof dbp);
    dbp.bucket_size = 8192;
    dbp.enzyme_present = 1;
    dbp.missed_cleavages = 1;
    dbp.min_len = -1;  /* None: EnzymeBuilder defaults (5 / 50) */
    dbp.max_len = -1;
    dbp.cleave_at = "KR";
    dbp.restrict_ = "P";
    dbp.c_terminal = -1;
    dbp.semi_enzymatic = -1;
    dbp.peptide_min_mass = 500.0f;
    dbp.peptide_max_mass = 5000.0f;
    dbp.ion_kinds = kinds;
    dbp.n_ion_kinds = 2;
    dbp.min_ion_index = 2;
    dbp.static_mod_keys = static_keys;
    dbp.static_mod_masses = static_masses;
    dbp.n_static_mods = 1;
    dbp.max_variable_mods = 2;
    dbp.decoy_tag = "rev_";
    dbp.generate_decoys = 1;
    dbp.peptides_only = 1;
    SageHostDb* host = NULL;
    CHECK(sage_hip_hostdb_build(fasta, &dbp, &host));
    SageDbView view;
    CHECK(sage_hip_hostdb_view(host, &view));
    SageDeviceDb* dev = NULL;
    CHECK(sage_hip_db_create(&view, 0, &dev));

    /* the Scorer struct literal of runner.rs:492-508 with input.rs:355-385 defaults */
    SageScorerParams sp;
    memset(&sp, 0, sizeof sp);
    sp.precursor_tol.kind = SAGE_TOL_PPM; sp.precursor_tol.lo = -20.0f; sp.precursor_tol.hi = 20.0f;
    sp.fragment_tol.kind = SAGE_TOL_PPM;  sp.fragment_tol.lo = -10.0f;  sp.fragment_tol.hi = 10.0f;
    sp.min_matched_peaks = 4;
    sp.min_isotope_err = 0; sp.max_isotope_err = 0;
    sp.min_precursor_charge = 2; sp.max_precursor_charge = 4;
    sp.max_fragment_charge = -1;
    sp.report_psms = 1;
    sp.score_type = 0;
    SageScorer* scorer = NULL;
    CHECK(sage_hip_scorer_create(dev, &sp, &scorer));

    SageMzml* run = NULL;
    CHECK(sage_hip_mzml_read(argv[2], 0, 2, &run));
    CHECK(sage_hip_mzml_check_searchable(run));
    SageRawBatch raw;
    CHECK(sage_hip_mzml_view(run, &raw));
    const uint32_t n = raw.n_spectra;
    SageDeviceBatch* batch = NULL;
    CHECK(sage_hip_batch_process_upload(scorer, &raw, 150, 1, 0.0f, 15, &batch, NULL));
    SageFeature* feats = (SageFeature*)calloc(n ? n : 1, sizeof(SageFeature));
    uint32_t* counts = (uint32_t*)calloc(n ? n : 1, sizeof(uint32_t));
    CHECK(sage_hip_score_resident(scorer, batch, feats, counts));

    /* the PSMs in (spectrum, rank) order, as Scorer::score results are collected (runner.rs:325) */
    uint64_t n_psm = 0;
    for (uint32_t i = 0; i < n; i++) n_psm += counts[i];
    SageFeature* flat = (SageFeature*)calloc(n_psm ? n_psm : 1, sizeof(SageFeature));
    const char** spec_ids = (const char**)calloc(n_psm ? n_psm : 1, sizeof(char*));
    uint64_t* psm_id = (uint64_t*)calloc(n_psm ? n_psm : 1, sizeof(uint64_t));
    uint32_t* pep_idx = (uint32_t*)calloc(n_psm ? n_psm : 1, sizeof(uint32_t));
    uint64_t k = 0;
    for (uint32_t i = 0; i < n; i++)
        for (uint32_t r = 0; r < counts[i]; r++) {
            flat[k] = feats[(size_t)i * sp.report_psms + r];
            flat[k].file_id = 0;
            spec_ids[k] = sage_hip_mzml_spectrum_id(run, i);
            psm_id[k] = k + 1;  /* PSM_COUNTER starts at 1 (scoring.rs:163) */
            pep_idx[k] = flat[k].peptide_idx;
            k++;
        }

    /* runner.rs:536-541 (predict_rt off): spectrum_fdr + picked_peptide + picked_protein on the device */
    uint32_t *pk = (uint32_t*)calloc(n_psm ? n_psm : 1, 4), *prk = (uint32_t*)calloc(n_psm ? n_psm : 1, 4), npk = 0, npr = 0;
    CHECK(sage_hip_hostdb_competition_keys(host, pep_idx, n_psm, pk, &npk, prk, &npr));
    float* cols = (float*)calloc((n_psm ? n_psm : 1) * 5, sizeof(float));
    uint32_t* order32 = (uint32_t*)calloc(n_psm ? n_psm : 1, 4);
    SageRescoreInput rin;
    memset(&rin, 0, sizeof rin);
    rin.n = n_psm;
    rin.features = flat;
    rin.precursor_tol = sp.precursor_tol;
    rin.peptide_key = pk; rin.n_peptide_keys = npk;
    rin.protein_key = prk; rin.n_protein_keys = npr;
    SageRescoreOutput rout;
    memset(&rout, 0, sizeof rout);
    rout.discriminant_score = cols;
    rout.posterior_error = cols + n_psm;
    rout.spectrum_q = cols + 2 * n_psm;
    rout.peptide_q = cols + 3 * n_psm;
    rout.protein_q = cols + 4 * n_psm;
    rout.order = order32;
    CHECK(sage_hip_rescore(0, &rin, &rout));
    uint64_t* order = (uint64_t*)calloc(n_psm ? n_psm : 1, 8);
    for (uint64_t j = 0; j < n_psm; j++) order[j] = order32[j];

    SagePostColumns post;
    memset(&post, 0, sizeof post);
    post.discriminant_score = rout.discriminant_score;
    post.posterior_error = rout.posterior_error;
    post.spectrum_q = rout.spectrum_q;
    post.peptide_q = rout.peptide_q;
    post.protein_q = rout.protein_q;
    const char* base = strrchr(argv[2], '/');
    const char* filenames[1] = {base ? base + 1 : argv[2]};
    CHECK(sage_hip_write_results(argv[3], SAGE_FORMAT_TSV, host, flat, n_psm, order, psm_id, filenames, 1, spec_ids, &post));
    printf("%u spectra, %llu PSMs, %llu at 1%% FDR, lda_fitted=%d\n", n, (unsigned long long)n_psm,
           (unsigned long long)rout.passing_spectrum, rout.lda_fitted);

    sage_hip_batch_free(batch);
    sage_hip_mzml_free(run);
    sage_hip_scorer_destroy(scorer);
    sage_hip_db_destroy(dev);
    sage_hip_hostdb_free(host);
    return 0;
}
