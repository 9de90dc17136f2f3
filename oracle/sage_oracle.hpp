// =============================================================================
// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product.
//
// A sequential, strict-IEEE C++ restatement of lazear/sage's fragment-index
// search-and-score path (crates/sage/src/{mass,modification,enzyme,fasta,
// peptide,ion_series,database,heap,spectrum,scoring}.rs @ v0.15.0-beta.2).
// Every function cites the reference file:line it follows.  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this code,
// and only as the checker / the CPU baseline — never as the thing shipped.
//
// Build with: g++ -O2 -std=c++17 -ffp-contract=off -fno-fast-math (see Makefile).
// The reference cannot be compiled here (Rust, no toolchain), so parity is
// pinned by the reference's own known-answer tests (oracle/selftest.cpp and
// tests/test_oracle_golden.py): see DESIGN.md "Oracle".
// =============================================================================
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <optional>
#include <string>
#include <utility>
#include <vector>

namespace sage_oracle {

// ---- mass.rs:5-8 -----------------------------------------------------------
constexpr float H2O = 18.010565f;
constexpr float PROTON = 1.0072764f;
constexpr float NEUTRON = 1.00335f;

// f32::total_cmp / f64::total_cmp (IEEE totalOrder): -1, 0, +1
int total_cmp(float a, float b);
int total_cmp(double a, double b);

// ---- mass.rs:10-57 ---------------------------------------------------------
struct Tolerance {
    enum Kind : int { PPM = 0, PCT = 1, DA = 2 };
    Kind kind = PPM;
    float lo = 0.f, hi = 0.f;
    static Tolerance Ppm(float l, float h) { return {PPM, l, h}; }
    static Tolerance Pct(float l, float h) { return {PCT, l, h}; }
    static Tolerance Da(float l, float h) { return {DA, l, h}; }
    std::pair<float, float> bounds(float center) const;  // mass.rs:21-35
    bool contains(float center, float rhs) const;        // mass.rs:37-40
    static float ppm_to_delta_mass(float center, float ppm) {  // mass.rs:42-44
        return ppm * center / 1000000.0f;
    }
    Tolerance scaled(float rhs) const;  // impl Mul<f32>, mass.rs:47-57
};

float monoisotopic(uint8_t aa);  // mass.rs:64-76
bool valid_aa(uint8_t aa);       // mass.rs:59-62

// ---- ion_series.rs:6-15 ----------------------------------------------------
enum class Kind : uint8_t { A = 0, B = 1, C = 2, X = 3, Y = 4, Z = 5 };
inline bool is_nterm_kind(Kind k) { return k == Kind::A || k == Kind::B || k == Kind::C; }

// ---- modification.rs:10-17 -------------------------------------------------
struct ModSpec {
    enum Type : int { PeptideN = 0, PeptideC = 1, ProteinN = 2, ProteinC = 3, Residue = 4 };
    Type type = Residue;
    int residue = -1;  // -1 == None (for the four terminal kinds)
    bool operator==(const ModSpec& o) const { return type == o.type && residue == o.residue; }
    bool operator<(const ModSpec& o) const {
        return type != o.type ? type < o.type : residue < o.residue;
    }
};
// modification.rs:66-104; returns false on the three error kinds
bool parse_modspec(const std::string& s, ModSpec& out);

// ---- enzyme.rs -------------------------------------------------------------
enum class Position : uint8_t { Nterm = 0, Cterm = 1, Full = 2, Internal = 3 };  // enzyme.rs:64-71

struct Digest {  // enzyme.rs:13-26
    bool decoy = false;
    bool semi_enzymatic = false;
    std::string sequence;
    std::string protein;
    uint8_t missed_cleavages = 0;
    Position position = Position::Internal;
};

struct DigestGroup {  // enzyme.rs:28-31
    Digest reference;
    std::vector<std::string> proteins;
};

struct DigestSite {  // enzyme.rs:125-133
    size_t start = 0, end = 0;
    uint8_t missed_cleavages = 0;
    bool semi_enzymatic = false;
};

struct Enzyme {  // enzyme.rs:113-123
    bool skip_suffix[26] = {};
    bool cleave[26] = {};  // the regex "[KR]" as a residue set
    bool dollar = false;   // the "$" no-cleavage enzyme
    bool c_terminal = true;
    bool semi_enzymatic = false;
    // enzyme.rs:135-187
    static std::optional<Enzyme> make(const std::string& cleave, const std::string& skip_suffix,
                                      bool c_terminal, bool semi_enzymatic);
    std::vector<DigestSite> cleavage_sites(const std::string& sequence) const;  // enzyme.rs:189-217
};

struct EnzymeParameters {  // enzyme.rs:103-111
    uint8_t missed_cleavages = 0;
    size_t min_len = 5, max_len = 50;
    std::optional<Enzyme> enzyme;
    std::vector<DigestSite> cleavage_sites(const std::string& sequence) const;  // enzyme.rs:221-240
    std::vector<Digest> digest(const std::string& sequence, const std::string& protein) const;  // :289-342
};

std::vector<DigestGroup> group_digests(std::vector<Digest> digests);  // enzyme.rs:33-62

// ---- fasta.rs --------------------------------------------------------------
struct Fasta {
    std::vector<std::pair<std::string, std::string>> targets;  // (accession, sequence)
    std::string decoy_tag;
    bool generate_decoys = true;
    static Fasta parse(const std::string& contents, const std::string& decoy_tag,
                       bool generate_decoys);                        // fasta.rs:16-56
    std::vector<Digest> digest(const EnzymeParameters& enzyme) const;  // fasta.rs:58-79
};

// ---- peptide.rs ------------------------------------------------------------
struct Peptide {  // peptide.rs:12-31
    bool decoy = false;
    std::string sequence;
    std::vector<float> modifications;
    std::optional<float> nterm, cterm;
    float monoisotopic = 0.f;
    uint8_t missed_cleavages = 0;
    bool semi_enzymatic = false;
    Position position = Position::Internal;
    std::vector<std::string> proteins;

    int initial_sort(const Peptide& other) const;  // peptide.rs:34-52
    int label() const { return decoy ? -1 : 1; }   // peptide.rs:74-79
    // peptide.rs:258-305
    std::vector<Peptide> apply(const std::vector<std::pair<ModSpec, float>>& variable_mods,
                               const std::vector<std::pair<ModSpec, float>>& static_mods,
                               size_t combinations) const;
    Peptide reverse() const;                            // peptide.rs:307-318
    static bool from_digest(const Digest& d, Peptide& out);  // TryFrom<Digest>, peptide.rs:357-388
    std::string to_string() const;                      // Display, peptide.rs:391-408
};

// ---- ion_series.rs:27-85 ---------------------------------------------------
// all L-1 cumulative neutral fragment masses of `kind`, in iterator order
std::vector<float> ion_series(const Peptide& p, Kind kind);

// ---- database.rs -----------------------------------------------------------
struct Theoretical {  // database.rs:378-382
    uint32_t peptide_index;
    float fragment_mz;
};

struct EnzymeBuilder {  // database.rs:15-57
    std::optional<uint8_t> missed_cleavages;
    std::optional<size_t> min_len, max_len;
    std::optional<std::string> cleave_at, restrict;
    std::optional<bool> c_terminal, semi_enzymatic;
    static EnzymeBuilder defaults();          // database.rs:29-41
    EnzymeParameters to_parameters() const;   // database.rs:43-57
};

struct IndexedDatabase;

struct Parameters {  // database.rs:122-139 (+ Builder::make_parameters :96-115 defaults)
    size_t bucket_size = 8192;
    EnzymeBuilder enzyme = EnzymeBuilder::defaults();
    float peptide_min_mass = 500.0f, peptide_max_mass = 5000.0f;
    std::vector<Kind> ion_kinds = {Kind::B, Kind::Y};
    size_t min_ion_index = 2;
    // HashMaps in the reference (iteration order random); here: deterministic vectors.
    std::vector<std::pair<ModSpec, float>> static_mods;
    std::vector<std::pair<ModSpec, std::vector<float>>> variable_mods;
    size_t max_variable_mods = 2;
    std::string decoy_tag = "rev_";
    bool generate_decoys = true;

    std::vector<Peptide> digest(const Fasta& fasta) const;                      // database.rs:162-219
    static void reorder_peptides(std::vector<Peptide>& target_decoys);          // database.rs:221-258
    IndexedDatabase build(const Fasta& fasta) const;                            // database.rs:260-263
    IndexedDatabase build_from_peptides(std::vector<Peptide> peptides) const;   // database.rs:265-364
};

struct IndexedQuery;

struct IndexedDatabase {  // database.rs:384-395
    std::vector<Peptide> peptides;
    std::vector<Theoretical> fragments;
    std::vector<Kind> ion_kinds;
    std::vector<float> min_value;
    size_t bucket_size = 8192;
    bool generate_decoys = true;
    std::string decoy_tag;
    IndexedQuery query(float precursor_mass, Tolerance precursor_tol, Tolerance fragment_tol) const;  // :402-425
};

// algorithmic-work counters: not in the reference; used to price the roofline
struct WorkCounters {
    uint64_t queries = 0;          // IndexedDatabase::query calls
    uint64_t page_searches = 0;    // (peak, fragment charge) probes
    uint64_t pages = 0;            // pages visited
    uint64_t scanned = 0;          // Theoretical entries inspected (slice[inner_left..inner_right])
    uint64_t hits = 0;             // entries passing the filter
    uint64_t peaks = 0;            // sum of P over spectra
    uint64_t rescored = 0;         // score_candidate calls
    uint64_t rescored_residues = 0;  // sum of L over score_candidate calls
    uint64_t reported = 0;         // features emitted
    uint64_t algorithmic_bytes = 0;  // closed form of SURVEY.md §8d
    void add(const WorkCounters& o);
};

struct IndexedQuery {  // database.rs:469-476
    const IndexedDatabase* db;
    float precursor_mass;
    Tolerance precursor_tol, fragment_tol;
    size_t pre_idx_lo, pre_idx_hi;
    // database.rs:480-536; calls f for every yielded fragment, in iterator order
    void page_search(float mass, const std::function<void(const Theoretical&)>& f,
                     WorkCounters* wc = nullptr) const;
};

// database.rs:549-561
template <class T, class S, class Key>
std::pair<size_t, size_t> binary_search_slice(const T* slice, size_t len, Key key, const S& low,
                                              const S& high) {
    // partition_point(|a| key(a, low) == Less)
    size_t lo = 0, hi = len;
    while (lo < hi) {
        size_t mid = lo + (hi - lo) / 2;
        if (key(slice[mid], low) < 0) lo = mid + 1; else hi = mid;
    }
    size_t left = lo == 0 ? 0 : lo - 1;  // saturating_sub(1)
    // slice[left..].partition_point(|a| key(a, high) != Greater) + left
    lo = left; hi = len;
    while (lo < hi) {
        size_t mid = lo + (hi - lo) / 2;
        if (key(slice[mid], high) <= 0) lo = mid + 1; else hi = mid;
    }
    return {left, lo};
}

// ---- heap.rs:7-60 ----------------------------------------------------------
// Less(a,b) is the strict `<` of the element type (PartialOrd::lt in the reference)
template <class T, class Less>
void sift_down(T* slice, size_t len, size_t index, Less less) {  // heap.rs:40-60
    while (index * 2 + 1 < len) {
        size_t smallest = index;
        if (less(slice[index * 2 + 1], slice[smallest])) smallest = index * 2 + 1;
        if (index * 2 + 2 < len) {
            if (less(slice[index * 2 + 2], slice[smallest])) smallest = index * 2 + 2;
        }
        if (smallest != index) {
            std::swap(slice[smallest], slice[index]);
            index = smallest;
        } else {
            break;
        }
    }
}
template <class T, class Less>
void bounded_min_heapify(T* slice, size_t len, size_t k, Less less) {  // heap.rs:7-28
    if (len <= k) return;
    for (size_t i = k / 2; i-- > 0;) sift_down(slice, k, i, less);
    for (size_t i = k; i < len; i++) {
        if (less(slice[0], slice[i])) {  // slice[i] > slice[0]
            std::swap(slice[i], slice[0]);
            sift_down(slice, k, 0, less);
        }
    }
}

// ---- spectrum.rs -----------------------------------------------------------
struct Precursor {  // spectrum.rs:46-55
    float mz = 0.f;
    std::optional<float> intensity;
    std::optional<uint8_t> charge;
    std::optional<Tolerance> isolation_window;
    std::optional<float> inverse_ion_mobility;
};

struct ProcessedSpectrum {  // spectrum.rs:57-79
    uint8_t level = 2;
    std::string id;
    size_t file_id = 0;
    float scan_start_time = 0.f;
    float ion_injection_time = 0.f;
    std::vector<Precursor> precursors;
    std::vector<float> masses, intensities, mobilities;
    float total_ion_current = 0.f;
};

struct RawSpectrum {  // spectrum.rs:81-106
    size_t file_id = 0;
    uint8_t ms_level = 2;
    std::string id;
    std::vector<Precursor> precursors;
    bool centroid = true;  // Representation
    float scan_start_time = 0.f, ion_injection_time = 0.f, total_ion_current = 0.f;
    std::vector<float> mz, intensity;
};

struct Deisotoped {  // spectrum.rs:27-37
    float mz, intensity;
    std::optional<uint8_t> charge;
    std::optional<size_t> envelope;
};

// spectrum.rs:134-159; returns -1 for None
long select_most_intense_peak(const float* masses, const float* intensities, size_t n, float center,
                              Tolerance tolerance, std::optional<float> offset);
std::vector<Deisotoped> deisotope(const float* mz, const float* inten, size_t n, uint8_t max_charge,
                                  float ppm, float min_mz);  // spectrum.rs:179-227
void path_compression(std::vector<Deisotoped>& peaks);         // spectrum.rs:230-239

struct SpectrumProcessor {  // spectrum.rs:39-44, 263-413
    size_t take_top_n = 150;
    float min_deisotope_mz = 0.f;
    bool deisotope = true;
    ProcessedSpectrum process(const RawSpectrum& s) const;
};

// ---- scoring.rs ------------------------------------------------------------
enum class ScoreType : int { SageHyperScore = 0, OpenMSHyperScore = 1 };

double lnfact(uint16_t n);  // scoring.rs:170-177
// f64::ln (sage_oracle.cpp): mode 0 = the platform libm, 1 = correctly rounded through libquadmath
void set_log_mode(int mode);
int get_log_mode();
double ln(double x);
double ln_correctly_rounded(double x);
double score_type_score(ScoreType t, uint16_t matched_b, uint16_t matched_y, float summed_b,
                        float summed_y);  // scoring.rs:179-201
uint8_t max_fragment_charge(std::optional<uint8_t> max_fragment_charge, uint8_t precursor_charge);  // :239-247

struct Run {  // scoring.rs:771-793
    size_t start = 0, length = 0, last = 0, longest = 0;
    void matched(size_t index);
};

struct Fragments {  // scoring.rs:152-161
    std::vector<int32_t> charges;
    std::vector<Kind> kinds;
    std::vector<int32_t> fragment_ordinals;
    std::vector<float> intensities, mz_calculated, mz_experimental;
};

struct Feature {  // scoring.rs:69-149 — the hot-path-owned fields
    uint32_t peptide_idx = 0xFFFFFFFFu;
    size_t peptide_len = 0;
    size_t spec_index = 0;  // position of the spectrum in the batch (spec_id stand-in)
    size_t file_id = 0;
    uint32_t rank = 0;
    int32_t label = 0;
    float expmass = 0, calcmass = 0;
    uint8_t charge = 0;
    float rt = 0, ims = 0;
    float delta_mass = 0, isotope_error = 0, average_ppm = 0;
    double hyperscore = 0, delta_next = 0, delta_best = 0;
    uint32_t matched_peaks = 0, longest_b = 0, longest_y = 0;
    float longest_y_pct = 0;
    uint8_t missed_cleavages = 0;
    float matched_intensity_pct = 0;
    uint32_t scored_candidates = 0;
    double poisson = 0;
    float ms2_intensity = 0;
    std::optional<Fragments> fragments;
};

struct PreScore {  // scoring.rs:43-49; derived Ord = lexicographic in field order
    uint16_t matched = 0;
    uint32_t peptide = 0xFFFFFFFFu;  // PeptideIx::default() == u32::MAX (database.rs:372-376)
    uint8_t precursor_charge = 0;
    int8_t isotope_error = 0;
};
bool prescore_less(const PreScore& a, const PreScore& b);

struct InitialHits {  // scoring.rs:52-67
    size_t matched_peaks = 0;
    size_t scored_candidates = 0;
    std::vector<PreScore> preliminary;
    void add_assign(InitialHits&& rhs);
};

struct Score {  // scoring.rs:17-30
    uint32_t peptide = 0xFFFFFFFFu;
    uint16_t matched_b = 0, matched_y = 0;
    float summed_b = 0, summed_y = 0;
    size_t longest_b = 0, longest_y = 0;
    double hyperscore = 0;
    float ppm_difference = 0;
    uint8_t precursor_charge = 0;
    int8_t isotope_error = 0;
};

struct Scorer {  // scoring.rs:210-232
    const IndexedDatabase* db = nullptr;
    Tolerance precursor_tol, fragment_tol;
    uint16_t min_matched_peaks = 4;
    int8_t min_isotope_err = 0, max_isotope_err = 0;
    uint8_t min_precursor_charge = 2, max_precursor_charge = 4;
    bool override_precursor_charge = false;
    std::optional<uint8_t> max_fragment_charge;
    bool chimera = false;
    size_t report_psms = 1;
    bool wide_window = false;
    bool annotate_matches = false;
    ScoreType score_type = ScoreType::SageHyperScore;
    mutable WorkCounters* wc = nullptr;  // instrumentation only

    std::vector<Feature> score(const ProcessedSpectrum& query) const;            // :300-309
    std::vector<Feature> score_standard(const ProcessedSpectrum& query) const;   // :465-474
    std::vector<Feature> score_chimera_fast(const ProcessedSpectrum& query) const;  // :648-672
    // :255-298 (keep[] is a plain byte vector here)
    void quick_score(const ProcessedSpectrum& query, bool prefilter_low_memory,
                     std::vector<uint8_t>& keep) const;

    // internals, exposed for tests
    void trim_hits(InitialHits& hits) const;                                      // :322-329
    InitialHits matched_peaks_with_isotope(const ProcessedSpectrum& query, float precursor_mass,
                                           uint8_t precursor_charge, Tolerance precursor_tol,
                                           int8_t isotope_error) const;           // :335-382
    InitialHits matched_peaks(const ProcessedSpectrum& query, float precursor_mass,
                              uint8_t precursor_charge, Tolerance precursor_tol) const;  // :384-416
    InitialHits initial_hits(const ProcessedSpectrum& query, const Precursor& precursor) const;  // :418-462
    void build_features(const ProcessedSpectrum& query, const Precursor& precursor,
                        const InitialHits& hits, size_t report_psms,
                        std::vector<Feature>& features) const;                    // :478-595
    void remove_matched_peaks(ProcessedSpectrum& query, const Feature& psm) const;  // :598-644
    std::pair<Score, std::optional<Fragments>> score_candidate(const ProcessedSpectrum& query,
                                                               const PreScore& pre) const;  // :675-767
    // independent cross-check (not in the reference): score every peptide in the precursor
    // window without the fragment index and without k-select (SURVEY.md §8c iv)
    std::vector<Score> brute_force_scores(const ProcessedSpectrum& query, float precursor_mass,
                                          uint8_t charge, Tolerance precursor_tol,
                                          int8_t isotope_error) const;
};

}  // namespace sage_oracle
