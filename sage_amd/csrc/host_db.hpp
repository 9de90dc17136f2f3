// Host-side IndexedDatabase construction for libsage_hip (product code).
// Mirrors sage-core's Parameters::build (crates/sage/src/database.rs:260-364) and everything it
// calls: fasta.rs, enzyme.rs, peptide.rs, modification.rs, ion_series.rs.  Output is the flat
// SoA/AoS layout that include/sage_hip.h's SageDbView exposes and that the device uploader consumes.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "../../include/sage_hip.h"

namespace sagehip {

// modification.rs:10-17 (ModificationSpecificity); residue < 0 == None
struct ModTarget {
    enum Where : uint8_t { PeptideN, PeptideC, ProteinN, ProteinC, Residue } where;
    int residue;
};
bool parse_mod_target(const char* key, ModTarget& out);  // modification.rs:66-104

struct DbBuildConfig {  // database.rs:122-139 after Builder::make_parameters (:96-115)
    uint64_t bucket_size = 8192;
    uint8_t missed_cleavages = 0;
    size_t min_len = 5, max_len = 50;
    std::string cleave_at = "KR", restrict_ = "P";
    bool c_terminal = true, semi_enzymatic = false;
    float peptide_min_mass = 500.0f, peptide_max_mass = 5000.0f;
    std::vector<uint8_t> ion_kinds = {SAGE_ION_B, SAGE_ION_Y};
    size_t min_ion_index = 2;
    std::vector<std::pair<ModTarget, float>> static_mods;
    std::vector<std::pair<ModTarget, float>> variable_mods;  // flattened (target, mass) pairs
    size_t max_variable_mods = 2;
    std::string decoy_tag = "rev_";
    bool generate_decoys = true;
    bool peptides_only = false;  // stop after reorder_peptides: the device generates the fragment index
};
DbBuildConfig config_from_params(const SageDbParams& p);

struct HostDb {
    std::vector<SageTheoretical> fragments;
    std::vector<float> min_value;
    uint64_t bucket_size = 8192;
    std::vector<float> pep_mono;
    std::vector<uint64_t> seq_off;
    std::vector<uint8_t> seq;
    std::vector<float> mods;
    std::vector<float> nterm, cterm;
    std::vector<uint8_t> decoy, missed, semi, position;  // position: enzyme.rs:64-71 (Nterm, Cterm, Full, Internal)
    std::vector<uint8_t> ion_kinds;
    // protein bookkeeping (not used by the scoring path; kept so writers can be added later)
    std::vector<std::string> protein_names;
    std::vector<uint64_t> pep_protein_off;
    std::vector<uint32_t> pep_protein_ids;
    std::string decoy_tag;
    bool generate_decoys = true;
    uint64_t min_ion_index = 2;

    uint64_t n_peptides() const { return pep_mono.size(); }
    SageDbView view() const;
    std::string peptide_string(uint64_t i) const;
    std::string peptide_proteins(uint64_t i) const;
    void competition_keys(const uint32_t* peptide_idx, uint64_t n, uint32_t* peptide_key, uint32_t& n_peptide_keys,
                          uint32_t* protein_key, uint32_t& n_protein_keys) const;
};

unsigned host_threads();
void parallel_for(size_t n, size_t grain, const std::function<void(size_t, size_t, unsigned)>& f);

HostDb build_database(const std::string& fasta_text, const DbBuildConfig& cfg, uint64_t first_target = 0,
                      uint64_t n_targets = ~0ull);
// the `prefilter` flow of sage-cli (runner.rs:104-127, :143-238)
uint64_t fasta_num_targets(const std::string& fasta_text, const DbBuildConfig& cfg);
uint64_t prefilter_chunk_size(const std::string& fasta_text, const DbBuildConfig& cfg, uint64_t requested);
HostDb merge_kept(const std::vector<const HostDb*>& chunks, const std::vector<const uint8_t*>& keep, const DbBuildConfig& cfg);

// mzml_reader.cpp: the MSn spectra of one mzML file as flat arrays (the layout of SageRawBatch) + their ids
struct MzmlRun {
    std::vector<uint64_t> peak_off;  // [n + 1]
    std::vector<float> mz, intensities;
    std::vector<float> precursor_mz, isolation_lo, isolation_hi, scan_start_time, inverse_ion_mobility;  // NaN == None
    std::vector<uint8_t> precursor_charge;                                                                 // 0 == None
    std::vector<uint8_t> centroid, has_precursor, ms_level;  // Representation::Centroid seen; a <precursor> with a selected ion; level
    std::vector<uint32_t> file_id;
    std::string ids;                 // NUL-separated spectrum ids
    std::vector<uint64_t> id_off;    // [n + 1] into ids
    uint64_t n() const { return precursor_mz.size(); }
};
bool read_mzml(const char* path, uint32_t file_id, int ms_level, MzmlRun& run, std::string& err);

// writers.cpp
bool write_results(const char* path, int format, const HostDb& db, const SageFeature* f, uint64_t n, const uint64_t* order,
                   const uint64_t* psm_id, const char* const* filenames, uint32_t n_files, const char* const* spec_ids,
                   const SagePostColumns* post, std::string& err);

// f32 residue masses, mass.rs:64-76
float residue_mass(uint8_t aa);
// IonSeries (ion_series.rs:36-85) for a flat peptide record; writes L-1 masses to out
void ion_series_flat(const uint8_t* seq, const float* mods, size_t len, float nterm /*NaN none*/, float mono,
                     uint8_t kind, float* out);

// SpectrumProcessor::process for one MS2 spectrum (spectrum.rs:279-412)
uint64_t process_ms2(uint64_t take_top_n, bool deisotope, float min_deisotope_mz, const float* mz,
                     const float* intensity, uint64_t n, uint8_t precursor_charge, float* out_mass,
                     float* out_intensity, float* out_tic);

}  // namespace sagehip
