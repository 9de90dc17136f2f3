// ORACLE — TEST INFRASTRUCTURE ONLY.  C entry points (ctypes) over sage_oracle.
// Struct layouts deliberately mirror include/sage_hip.h so the Python test harness can feed the
// same buffers to the oracle and to the product, but nothing here is linked into the product.
#include <omp.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <string>

#include "sage_oracle.hpp"

using namespace sage_oracle;

extern "C" {

struct OrcTolerance {
    int32_t kind;  // 0 ppm, 1 pct, 2 da
    float lo, hi;
};

struct OrcDbParams {
    uint64_t bucket_size;
    int32_t missed_cleavages;  // -1 = None (=> 1, database.rs:46)
    int32_t min_len, max_len;  // -1 = None
    const char* cleave_at;     // NULL = None
    const char* restrict_;     // NULL = None
    int32_t c_terminal;        // -1 None
    int32_t semi_enzymatic;    // -1 None
    int32_t enzyme_present;    // 0 => EnzymeBuilder::default() (database.rs:105)
    float peptide_min_mass, peptide_max_mass;
    const uint8_t* ion_kinds;  // 0..5 = a,b,c,x,y,z
    uint32_t n_ion_kinds;
    uint64_t min_ion_index;
    const char* const* static_mod_keys;
    const float* static_mod_masses;
    uint32_t n_static_mods;
    const char* const* var_mod_keys;  // one entry per (key, mass) pair, in config order
    const float* var_mod_masses;
    uint32_t n_var_mods;
    uint64_t max_variable_mods;
    const char* decoy_tag;
    int32_t generate_decoys;
};

struct OrcScorerParams {
    OrcTolerance precursor_tol, fragment_tol;
    uint16_t min_matched_peaks;
    int8_t min_isotope_err, max_isotope_err;
    uint8_t min_precursor_charge, max_precursor_charge;
    uint8_t override_precursor_charge;
    uint8_t chimera;
    int16_t max_fragment_charge;  // -1 = None
    uint8_t wide_window;
    uint8_t annotate_matches;
    uint32_t report_psms;
    int32_t score_type;
};

struct OrcSpectrumBatch {
    uint32_t n_spectra;
    const uint64_t* peak_off;
    const float* masses;
    const float* intensities;
    const float* precursor_mz;
    const uint8_t* precursor_charge;  // 0 = None
    const float* isolation_lo;        // NaN = None; Tolerance::Da(lo, hi)
    const float* isolation_hi;
    const float* total_ion_current;
    const float* scan_start_time;
    const float* inverse_ion_mobility;  // NaN = None
    const uint32_t* file_id;
};

struct OrcFeature {
    uint32_t spec_index, peptide_idx, rank;
    int32_t label;
    float expmass, calcmass, rt, ims, delta_mass, isotope_error, average_ppm;
    float longest_y_pct, matched_intensity_pct, ms2_intensity;
    double hyperscore, delta_next, delta_best, poisson;
    uint32_t matched_peaks, longest_b, longest_y, scored_candidates;
    uint32_t peptide_len, file_id;
    uint8_t charge, missed_cleavages;
    uint8_t pad[6];
};

struct OrcWork {
    uint64_t queries, page_searches, pages, scanned, hits, peaks, rescored, rescored_residues, reported,
        algorithmic_bytes;
};

static Tolerance tol(const OrcTolerance& t) { return Tolerance{(Tolerance::Kind)t.kind, t.lo, t.hi}; }

static Parameters make_params(const OrcDbParams* p) {
    Parameters P;
    // Builder::make_parameters, database.rs:96-115
    uint64_t b = p->bucket_size ? p->bucket_size : 8192;
    uint64_t pow2 = 1;
    while (pow2 < b) pow2 <<= 1;
    P.bucket_size = pow2;
    if (p->enzyme_present) {
        EnzymeBuilder e;
        if (p->missed_cleavages >= 0) e.missed_cleavages = (uint8_t)p->missed_cleavages;
        if (p->min_len >= 0) e.min_len = (size_t)p->min_len;
        if (p->max_len >= 0) e.max_len = (size_t)p->max_len;
        if (p->cleave_at) e.cleave_at = std::string(p->cleave_at);
        if (p->restrict_) e.restrict = std::string(p->restrict_);
        if (p->c_terminal >= 0) e.c_terminal = p->c_terminal != 0;
        if (p->semi_enzymatic >= 0) e.semi_enzymatic = p->semi_enzymatic != 0;
        P.enzyme = e;
    }
    P.peptide_min_mass = p->peptide_min_mass;
    P.peptide_max_mass = p->peptide_max_mass;
    P.ion_kinds.clear();
    for (uint32_t i = 0; i < p->n_ion_kinds; i++) P.ion_kinds.push_back((Kind)p->ion_kinds[i]);
    P.min_ion_index = p->min_ion_index;
    for (uint32_t i = 0; i < p->n_static_mods; i++) {
        ModSpec m;
        if (parse_modspec(p->static_mod_keys[i], m)) {
            bool dup = false;
            for (auto& sm : P.static_mods)
                if (sm.first == m) { sm.second = p->static_mod_masses[i]; dup = true; }
            if (!dup) P.static_mods.emplace_back(m, p->static_mod_masses[i]);
        }
    }
    std::sort(P.static_mods.begin(), P.static_mods.end(),
              [](const auto& a, const auto& b) { return a.first < b.first; });
    for (uint32_t i = 0; i < p->n_var_mods; i++) {
        ModSpec m;
        if (parse_modspec(p->var_mod_keys[i], m)) {
            bool dup = false;
            for (auto& vm : P.variable_mods)
                if (vm.first == m) { vm.second.push_back(p->var_mod_masses[i]); dup = true; }
            if (!dup) P.variable_mods.push_back({m, {p->var_mod_masses[i]}});
        }
    }
    P.max_variable_mods = std::max<uint64_t>(p->max_variable_mods, 1);  // .map(|x| x.max(1))
    P.decoy_tag = p->decoy_tag ? p->decoy_tag : "rev_";
    P.generate_decoys = p->generate_decoys != 0;
    return P;
}

void* orc_db_build(const char* fasta_text, const OrcDbParams* p) {
    Parameters P = make_params(p);
    Fasta fasta = Fasta::parse(fasta_text, P.decoy_tag, P.generate_decoys);
    return new IndexedDatabase(P.build(fasta));
}

// ---- the `prefilter` flow (sage-cli runner.rs:104-127, :143-238) ----
// one chunk of Fasta::iter_chunks (fasta.rs:81-89): targets [first, first + count)
void* orc_db_build_chunk(const char* fasta_text, const OrcDbParams* p, uint64_t first, uint64_t count) {
    Parameters P = make_params(p);
    Fasta fasta = Fasta::parse(fasta_text, P.decoy_tag, P.generate_decoys);
    Fasta chunk;
    chunk.decoy_tag = fasta.decoy_tag;
    chunk.generate_decoys = fasta.generate_decoys;
    for (uint64_t i = first; i < fasta.targets.size() && i < first + count; i++) chunk.targets.push_back(fasta.targets[i]);
    return new IndexedDatabase(P.build(chunk));
}
uint64_t orc_fasta_num_targets(const char* fasta_text, const OrcDbParams* p) {
    Parameters P = make_params(p);
    return Fasta::parse(fasta_text, P.decoy_tag, P.generate_decoys).targets.size();
}
// Parameters::auto_calculate_prefilter_chunk_size (database.rs:142-160)
uint64_t orc_prefilter_chunk_size(const char* fasta_text, const OrcDbParams* p, uint64_t requested) {
    if (requested) return requested;
    Parameters P = make_params(p);
    Fasta fasta = Fasta::parse(fasta_text, P.decoy_tag, P.generate_decoys);
    const uint64_t max_peps_per_chunk = 1ull << 23;
    const uint64_t total_unmodified_pep_count = fasta.digest(P.enzyme.to_parameters()).size();
    const uint64_t mod_count_estimate = (P.variable_mods.size() + 1) * (1ull << P.max_variable_mods);
    const uint64_t chunk_count = mod_count_estimate * total_unmodified_pep_count / max_peps_per_chunk;
    return chunk_count == 0 ? fasta.targets.size() : fasta.targets.size() / chunk_count;
}
// runner.rs:215-238: retain keep[ix] peptides of every chunk database, reorder_peptides, build_from_peptides
void* orc_db_merge_kept(void* const* chunks, const uint8_t* const* keep, uint32_t n_chunks, const OrcDbParams* p) {
    Parameters P = make_params(p);
    std::vector<Peptide> all;
    for (uint32_t c = 0; c < n_chunks; c++) {
        const IndexedDatabase* db = (const IndexedDatabase*)chunks[c];
        for (size_t i = 0; i < db->peptides.size(); i++)
            if (keep[c][i]) all.push_back(db->peptides[i]);
    }
    Parameters::reorder_peptides(all);
    return new IndexedDatabase(P.build_from_peptides(std::move(all)));
}

// Construct an oracle database from a flat (product-shaped) index: used by the cpu_baseline leg
// so both legs score against byte-identical inputs.
void* orc_db_from_arrays(const uint32_t* frag_pep, const float* frag_mz, uint64_t nf, const float* min_value,
                         uint64_t nb, uint64_t bucket_size, const float* pep_mono, const uint64_t* seq_off,
                         const uint8_t* seq, const float* mods, const float* nterm, const float* cterm,
                         const uint8_t* decoy, const uint8_t* missed, uint64_t np, const uint8_t* ion_kinds,
                         uint32_t n_kinds) {
    auto* db = new IndexedDatabase();
    db->fragments.resize(nf);
    for (uint64_t i = 0; i < nf; i++) db->fragments[i] = {frag_pep[i], frag_mz[i]};
    db->min_value.assign(min_value, min_value + nb);
    db->bucket_size = bucket_size;
    db->peptides.resize(np);
    for (uint64_t i = 0; i < np; i++) {
        Peptide& p = db->peptides[i];
        p.sequence.assign((const char*)seq + seq_off[i], seq_off[i + 1] - seq_off[i]);
        p.modifications.assign(mods + seq_off[i], mods + seq_off[i + 1]);
        if (!std::isnan(nterm[i])) p.nterm = nterm[i];
        if (cterm && !std::isnan(cterm[i])) p.cterm = cterm[i];
        p.monoisotopic = pep_mono[i];
        p.decoy = decoy[i] != 0;
        p.missed_cleavages = missed[i];
    }
    for (uint32_t i = 0; i < n_kinds; i++) db->ion_kinds.push_back((Kind)ion_kinds[i]);
    return db;
}

void orc_db_free(void* h) { delete (IndexedDatabase*)h; }
uint64_t orc_db_num_peptides(void* h) { return ((IndexedDatabase*)h)->peptides.size(); }
uint64_t orc_db_num_fragments(void* h) { return ((IndexedDatabase*)h)->fragments.size(); }
uint64_t orc_db_num_buckets(void* h) { return ((IndexedDatabase*)h)->min_value.size(); }
uint64_t orc_db_bucket_size(void* h) { return ((IndexedDatabase*)h)->bucket_size; }
uint64_t orc_db_total_residues(void* h) {
    uint64_t n = 0;
    for (auto& p : ((IndexedDatabase*)h)->peptides) n += p.sequence.size();
    return n;
}
void orc_db_copy_fragments(void* h, uint32_t* pep, float* mz) {
    auto* db = (IndexedDatabase*)h;
    for (size_t i = 0; i < db->fragments.size(); i++) {
        pep[i] = db->fragments[i].peptide_index;
        mz[i] = db->fragments[i].fragment_mz;
    }
}
void orc_db_copy_min_value(void* h, float* out) {
    auto* db = (IndexedDatabase*)h;
    std::memcpy(out, db->min_value.data(), db->min_value.size() * 4);
}
void orc_db_copy_peptides(void* h, float* mono, uint8_t* decoy, uint8_t* missed, float* nterm, float* cterm,
                          uint64_t* seq_off, uint8_t* seq, float* mods) {
    auto* db = (IndexedDatabase*)h;
    uint64_t off = 0;
    for (size_t i = 0; i < db->peptides.size(); i++) {
        const Peptide& p = db->peptides[i];
        mono[i] = p.monoisotopic;
        decoy[i] = p.decoy;
        missed[i] = p.missed_cleavages;
        nterm[i] = p.nterm ? *p.nterm : NAN;
        cterm[i] = p.cterm ? *p.cterm : NAN;
        seq_off[i] = off;
        std::memcpy(seq + off, p.sequence.data(), p.sequence.size());
        std::memcpy(mods + off, p.modifications.data(), p.modifications.size() * 4);
        off += p.sequence.size();
    }
    seq_off[db->peptides.size()] = off;
}
// "[+42]-MEWK..." display strings, '\n' separated; returns required size
uint64_t orc_db_peptide_strings(void* h, char* out, uint64_t cap) {
    auto* db = (IndexedDatabase*)h;
    std::string s;
    for (auto& p : db->peptides) { s += p.to_string(); s += '\n'; }
    if (out && cap >= s.size() + 1) std::memcpy(out, s.c_str(), s.size() + 1);
    return s.size() + 1;
}
// proteins of peptide i joined by ';'
uint64_t orc_db_peptide_proteins(void* h, uint64_t i, char* out, uint64_t cap) {
    auto* db = (IndexedDatabase*)h;
    std::string s;
    for (size_t j = 0; j < db->peptides[i].proteins.size(); j++) {
        if (j) s += ';';
        s += db->peptides[i].proteins[j];
    }
    if (out && cap >= s.size() + 1) std::memcpy(out, s.c_str(), s.size() + 1);
    return s.size() + 1;
}

// IndexedDatabase::query + page_search (database.rs:402-425, 480-536): visited fragment slots
// are returned as indices into db.fragments
uint64_t orc_db_page_search(void* h, float precursor_mass, OrcTolerance ptol, OrcTolerance ftol, float mass,
                            uint64_t* out_idx, uint64_t cap, uint64_t* pre_lo, uint64_t* pre_hi) {
    auto* db = (IndexedDatabase*)h;
    IndexedQuery q = db->query(precursor_mass, tol(ptol), tol(ftol));
    if (pre_lo) *pre_lo = q.pre_idx_lo;
    if (pre_hi) *pre_hi = q.pre_idx_hi;
    uint64_t n = 0;
    q.page_search(mass, [&](const Theoretical& f) {
        if (n < cap) out_idx[n] = (uint64_t)(&f - db->fragments.data());
        n++;
    });
    return n;
}

// SpectrumProcessor::process for one MS2 spectrum (spectrum.rs:338-412). out_* capacity >= n.
uint64_t orc_process_ms2(uint64_t take_top_n, int deisotope, float min_deisotope_mz, const float* mz,
                         const float* inten, uint64_t n, uint8_t precursor_charge /*0 = None*/,
                         float* out_mass, float* out_int, float* out_tic) {
    SpectrumProcessor sp;
    sp.take_top_n = take_top_n;
    sp.deisotope = deisotope != 0;
    sp.min_deisotope_mz = min_deisotope_mz;
    RawSpectrum raw;
    raw.ms_level = 2;
    raw.mz.assign(mz, mz + n);
    raw.intensity.assign(inten, inten + n);
    Precursor pr;
    if (precursor_charge) pr.charge = precursor_charge;
    raw.precursors.push_back(pr);
    ProcessedSpectrum out = sp.process(raw);
    std::memcpy(out_mass, out.masses.data(), out.masses.size() * 4);
    std::memcpy(out_int, out.intensities.data(), out.intensities.size() * 4);
    *out_tic = out.total_ion_current;
    return out.masses.size();
}

static Scorer make_scorer(const IndexedDatabase* db, const OrcScorerParams* sp) {
    Scorer s;
    s.db = db;
    s.precursor_tol = tol(sp->precursor_tol);
    s.fragment_tol = tol(sp->fragment_tol);
    s.min_matched_peaks = sp->min_matched_peaks;
    s.min_isotope_err = sp->min_isotope_err;
    s.max_isotope_err = sp->max_isotope_err;
    s.min_precursor_charge = sp->min_precursor_charge;
    s.max_precursor_charge = sp->max_precursor_charge;
    s.override_precursor_charge = sp->override_precursor_charge != 0;
    if (sp->max_fragment_charge >= 0) s.max_fragment_charge = (uint8_t)sp->max_fragment_charge;
    s.chimera = sp->chimera != 0;
    s.report_psms = sp->report_psms;
    s.wide_window = sp->wide_window != 0;
    s.annotate_matches = false;
    s.score_type = (ScoreType)sp->score_type;
    return s;
}

static ProcessedSpectrum make_spectrum(const OrcSpectrumBatch* b, uint32_t i) {
    ProcessedSpectrum q;
    q.level = 2;
    q.file_id = b->file_id ? b->file_id[i] : 0;
    q.scan_start_time = b->scan_start_time ? b->scan_start_time[i] : 0.0f;
    uint64_t s = b->peak_off[i], e = b->peak_off[i + 1];
    q.masses.assign(b->masses + s, b->masses + e);
    q.intensities.assign(b->intensities + s, b->intensities + e);
    q.total_ion_current = b->total_ion_current[i];
    Precursor pr;
    pr.mz = b->precursor_mz[i];
    if (b->precursor_charge && b->precursor_charge[i]) pr.charge = b->precursor_charge[i];
    if (b->isolation_lo && b->isolation_hi && !std::isnan(b->isolation_lo[i]) && !std::isnan(b->isolation_hi[i]))
        pr.isolation_window = Tolerance::Da(b->isolation_lo[i], b->isolation_hi[i]);
    if (b->inverse_ion_mobility && !std::isnan(b->inverse_ion_mobility[i]))
        pr.inverse_ion_mobility = b->inverse_ion_mobility[i];
    q.precursors.push_back(pr);
    return q;
}

static void to_c(const Feature& f, uint32_t spec_index, OrcFeature* o) {
    std::memset(o, 0, sizeof *o);
    o->spec_index = spec_index;
    o->peptide_idx = f.peptide_idx;
    o->rank = f.rank;
    o->label = f.label;
    o->expmass = f.expmass;
    o->calcmass = f.calcmass;
    o->rt = f.rt;
    o->ims = f.ims;
    o->delta_mass = f.delta_mass;
    o->isotope_error = f.isotope_error;
    o->average_ppm = f.average_ppm;
    o->longest_y_pct = f.longest_y_pct;
    o->matched_intensity_pct = f.matched_intensity_pct;
    o->ms2_intensity = f.ms2_intensity;
    o->hyperscore = f.hyperscore;
    o->delta_next = f.delta_next;
    o->delta_best = f.delta_best;
    o->poisson = f.poisson;
    o->matched_peaks = f.matched_peaks;
    o->longest_b = f.longest_b;
    o->longest_y = f.longest_y;
    o->scored_candidates = f.scored_candidates;
    o->peptide_len = (uint32_t)f.peptide_len;
    o->file_id = (uint32_t)f.file_id;
    o->charge = f.charge;
    o->missed_cleavages = f.missed_cleavages;
}

// Scorer::score over a batch, one task per spectrum, dynamic schedule (mirrors the rayon par_iter at
// sage-cli/src/runner.rs:311-325).  out: n_spectra * report_psms records; out_count[n_spectra].
// Returns elapsed milliseconds of the scoring loop (runner.rs:327-330 times the same region).
double orc_score_batch(void* h, const OrcScorerParams* sp, const OrcSpectrumBatch* batch, OrcFeature* out,
                       uint32_t* out_count, int nthreads, OrcWork* work) {
    auto* db = (IndexedDatabase*)h;
    Scorer scorer = make_scorer(db, sp);
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    std::vector<WorkCounters> wcs(nthreads);
    uint32_t n = batch->n_spectra;
    auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel num_threads(nthreads)
    {
        Scorer local = scorer;
        local.wc = work ? &wcs[omp_get_thread_num()] : nullptr;
#pragma omp for schedule(dynamic, 16)
        for (int64_t i = 0; i < (int64_t)n; i++) {
            ProcessedSpectrum q = make_spectrum(batch, (uint32_t)i);
            std::vector<Feature> feats = local.score(q);
            out_count[i] = (uint32_t)feats.size();
            for (size_t j = 0; j < feats.size() && j < sp->report_psms; j++)
                to_c(feats[j], (uint32_t)i, out + (size_t)i * sp->report_psms + j);
        }
    }
    auto t1 = std::chrono::steady_clock::now();
    if (work) {
        WorkCounters tot;
        for (auto& w : wcs) tot.add(w);
        *work = OrcWork{tot.queries, tot.page_searches, tot.pages, tot.scanned, tot.hits, tot.peaks,
                        tot.rescored, tot.rescored_residues, tot.reported, tot.algorithmic_bytes};
    }
    return std::chrono::duration<double, std::milli>(t1 - t0).count();
}

// initial_hits for one spectrum (scoring.rs:418-462): the trimmed preliminary list in heap-layout
// order.  Returns its length; out arrays need capacity >= max(50, 2*report_psms).
uint64_t orc_initial_hits(void* h, const OrcScorerParams* sp, const OrcSpectrumBatch* batch, uint32_t i,
                          uint16_t* matched, uint32_t* peptide, uint8_t* charge, int8_t* iso,
                          uint64_t cap, uint64_t* matched_peaks, uint64_t* scored_candidates) {
    auto* db = (IndexedDatabase*)h;
    Scorer scorer = make_scorer(db, sp);
    ProcessedSpectrum q = make_spectrum(batch, i);
    InitialHits hits = scorer.initial_hits(q, q.precursors.front());
    *matched_peaks = hits.matched_peaks;
    *scored_candidates = hits.scored_candidates;
    for (size_t j = 0; j < hits.preliminary.size() && j < cap; j++) {
        matched[j] = hits.preliminary[j].matched;
        peptide[j] = hits.preliminary[j].peptide;
        charge[j] = hits.preliminary[j].precursor_charge;
        iso[j] = hits.preliminary[j].isotope_error;
    }
    return hits.preliminary.size();
}

// brute-force cross-check for one (spectrum, charge, isotope): number of peptides in the window
// and their (peptide, matched_b+matched_y, hyperscore)
uint64_t orc_brute_force(void* h, const OrcScorerParams* sp, const OrcSpectrumBatch* batch, uint32_t i,
                         uint8_t charge, int8_t iso, uint32_t* peptide, uint32_t* matched, double* hyperscore,
                         uint64_t cap) {
    auto* db = (IndexedDatabase*)h;
    Scorer scorer = make_scorer(db, sp);
    ProcessedSpectrum q = make_spectrum(batch, i);
    float mass = (q.precursors.front().mz - PROTON) * (float)charge;
    auto v = scorer.brute_force_scores(q, mass, charge, scorer.precursor_tol, iso);
    for (size_t j = 0; j < v.size() && j < cap; j++) {
        peptide[j] = v[j].peptide;
        matched[j] = (uint32_t)v[j].matched_b + v[j].matched_y;
        hyperscore[j] = v[j].hyperscore;
    }
    return v.size();
}

// Scorer::score with annotate_matches = true (scoring.rs:739-752): the Fragments of every reported PSM, flattened.
// psm_off[i * report_psms + r] .. psm_off[i * report_psms + r + 1] delimit PSM r of spectrum i (empty for r >= count).
// Returns the total number of entries (arrays are filled up to `cap`).
uint64_t orc_annotate_batch(void* h, const OrcScorerParams* sp, const OrcSpectrumBatch* batch, uint64_t* psm_off,
                            uint8_t* kinds, int32_t* charges, int32_t* ordinals, float* intensities, float* mz_calculated,
                            float* mz_experimental, uint64_t cap) {
    auto* db = (IndexedDatabase*)h;
    Scorer scorer = make_scorer(db, sp);
    scorer.annotate_matches = true;
    uint64_t total = 0;
    const uint32_t n = batch->n_spectra, rp = sp->report_psms;
    for (uint32_t i = 0; i < n; i++) {
        ProcessedSpectrum q = make_spectrum(batch, i);
        std::vector<Feature> feats = scorer.score(q);
        for (uint32_t r = 0; r < rp; r++) {
            psm_off[(size_t)i * rp + r] = total;
            if (r >= feats.size() || !feats[r].fragments) continue;
            const Fragments& f = *feats[r].fragments;
            for (size_t j = 0; j < f.kinds.size(); j++, total++) {
                if (total >= cap) continue;
                kinds[total] = (uint8_t)f.kinds[j];
                charges[total] = f.charges[j];
                ordinals[total] = f.fragment_ordinals[j];
                intensities[total] = f.intensities[j];
                mz_calculated[total] = f.mz_calculated[j];
                mz_experimental[total] = f.mz_experimental[j];
            }
        }
    }
    psm_off[(size_t)n * rp] = total;
    return total;
}

// Scorer::quick_score (scoring.rs:255-298) over a batch: keep[peptide] |= identified
void orc_quick_score(void* h, const OrcScorerParams* sp, const OrcSpectrumBatch* batch, int prefilter_low_memory,
                     uint8_t* keep) {
    auto* db = (IndexedDatabase*)h;
    Scorer scorer = make_scorer(db, sp);
    std::vector<uint8_t> k(db->peptides.size(), 0);
    for (uint32_t i = 0; i < batch->n_spectra; i++) {
        ProcessedSpectrum q = make_spectrum(batch, i);
        scorer.quick_score(q, prefilter_low_memory != 0, k);
    }
    for (size_t i = 0; i < k.size(); i++) keep[i] |= k[i];
}

void orc_tol_bounds(OrcTolerance t, float center, float* lo, float* hi) {
    auto b = tol(t).bounds(center);
    *lo = b.first;
    *hi = b.second;
}
int orc_max_threads() { return omp_get_max_threads(); }

// f64::ln of the reference: 0 = the platform libm (default), 1 = correctly rounded (libquadmath) — sage_oracle.cpp
void orc_set_log_mode(int mode) { set_log_mode(mode); }
int orc_get_log_mode() { return get_log_mode(); }
void orc_ln_batch(int mode, const double* x, uint64_t n, double* out) {
    for (uint64_t i = 0; i < n; i++) out[i] = mode ? ln_correctly_rounded(x[i]) : std::log(x[i]);
}

}  // extern "C"
