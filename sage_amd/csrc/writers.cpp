// writers.cpp — results.sage.tsv / results.sage.pin rows on the host, in C++.
//
// Column order and number formatting follow the reference byte for byte: sage-cli/src/runner.rs:687-780 (serialize_feature),
// :830-905 (header), :938-1084 (serialize_pin), :1086-1135.  Integers as `itoa` prints them, floats as `ryu::Buffer::format`
// does: the shortest digits that round-trip (std::to_chars gives the same digits — both are exact shortest round-trip
// algorithms), laid out by ryu's pretty printer (ryu/src/pretty/mod.rs).  sage_amd/output.py holds the same logic in Python
// (kept for the small matched-fragments table and as the cross-check in tests/test_cli_io.py); this file exists because a
// Python loop over half a million PSMs x 43 columns is slower than the search that produced them.
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "host_db.hpp"

namespace sagehip {

namespace {

template <class F>
void ryu_append(std::string& out, F x) {
    constexpr bool is_f32 = sizeof(F) == 4;
    if (std::isnan(x)) {
        out += "NaN";
        return;
    }
    if (std::isinf(x)) {
        out += x > 0 ? "inf" : "-inf";
        return;
    }
    if (x == 0) {
        out += std::signbit(x) ? "-0.0" : "0.0";
        return;
    }
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof buf - 1, x, std::chars_format::scientific);  // [-]d[.ddd]e[+-]XX, shortest digits
    *r.ptr = '\0';
    const char* p = buf;
    if (*p == '-') {
        out += '-';
        ++p;
    }
    char digits[32];
    int len = 0;
    for (; p < r.ptr && *p != 'e'; ++p)
        if (*p != '.') digits[len++] = *p;
    const int e10 = std::atoi(p + 1);
    const int kk = e10 + 1;   // 10^(kk-1) <= |x| < 10^kk
    const int k = kk - len;   // x = digits * 10^k
    const int hi = is_f32 ? 13 : 16, lo = is_f32 ? -6 : -5;
    if (0 <= k && kk <= hi) {
        out.append(digits, len);
        out.append((size_t)k, '0');
        out += ".0";
    } else if (0 < kk && kk <= hi) {
        out.append(digits, kk);
        out += '.';
        out.append(digits + kk, len - kk);
    } else if (lo < kk && kk <= 0) {
        out += "0.";
        out.append((size_t)(-kk), '0');
        out.append(digits, len);
    } else if (len == 1) {
        out += digits[0];
        out += 'e';
        out += std::to_string(kk - 1);
    } else {
        out += digits[0];
        out += '.';
        out.append(digits + 1, len - 1);
        out += 'e';
        out += std::to_string(kk - 1);
    }
}

inline void f32(std::string& o, float x) {
    ryu_append<float>(o, x);
    o += '\t';
}
inline void f64(std::string& o, double x) {
    ryu_append<double>(o, x);
    o += '\t';
}
template <class I>
inline void itoa(std::string& o, I x) {
    o += std::to_string(x);
    o += '\t';
}
inline void str(std::string& o, const std::string& s) {
    o += s;
    o += '\t';
}
inline float col(const float* p, uint64_t i, float dflt) { return p ? p[i] : dflt; }

const char* const kTsvHeader =
    "psm_id\tpeptide\tproteins\tprotein_groups\tnum_proteins\tnum_protein_groups\tfilename\tscannr\trank\tlabel\texpmass\t"
    "calcmass\tcharge\tpeptide_len\tmissed_cleavages\tsemi_enzymatic\tisotope_error\tprecursor_ppm\tfragment_ppm\thyperscore\t"
    "delta_next\tdelta_best\trt\taligned_rt\tpredicted_rt\tdelta_rt_model\tion_mobility\tpredicted_mobility\tdelta_mobility\t"
    "matched_peaks\tlongest_b\tlongest_y\tlongest_y_pct\tmatched_intensity_pct\tscored_candidates\tpoisson\t"
    "sage_discriminant_score\tposterior_error\tspectrum_q\tpeptide_q\tprotein_q\tprotein_group_q\tms2_intensity\n";
const char* const kPinHeader =
    "SpecId\tLabel\tScanNr\tExpMass\tCalcMass\tFileName\tretentiontime\tion_mobility\trank\tz=2\tz=3\tz=4\tz=5\tz=6\tz=other\t"
    "peptide_len\tmissed_cleavages\tsemi_enzymatic\tisotope_error\tln(precursor_ppm)\tfragment_ppm\tln(hyperscore)\t"
    "ln(delta_next)\tln(delta_best)\taligned_rt\tpredicted_rt\tsqrt(delta_rt_model)\tpredicted_mobility\tsqrt(delta_mobility)\t"
    "matched_peaks\tlongest_b\tlongest_y\tlongest_y_pct\tln(matched_intensity_pct)\tscored_candidates\tln(-poisson)\t"
    "posterior_error\tPeptide\tProteins\n";

// the last capture of r"scan=(\d+)" in the spectrum id, or the whole id (runner.rs:944-948)
std::string scan_number(const char* id) {
    const char* best = nullptr;
    size_t best_len = 0;
    for (const char* p = id; (p = std::strstr(p, "scan=")) != nullptr; p += 5) {
        const char* d = p + 5;
        size_t n = 0;
        while (d[n] >= '0' && d[n] <= '9') ++n;
        if (n) {
            best = d;
            best_len = n;
        }
    }
    return best ? std::string(best, best_len) : std::string(id);
}

}  // namespace

// format: 0 = results.sage.tsv, 1 = results.sage.pin.  Returns false (errno set by fopen / fwrite) when the file cannot be
// written.  `order` (nullable) lists the row order; psm_id / spec_ids are indexed like `features`.
bool write_results(const char* path, int format, const HostDb& db, const SageFeature* f, uint64_t n, const uint64_t* order,
                   const uint64_t* psm_id, const char* const* filenames, uint32_t n_files, const char* const* spec_ids,
                   const SagePostColumns* post, std::string& err) {
    static const SagePostColumns kNone{};
    const SagePostColumns& pc = post ? *post : kNone;
    FILE* fh = std::fopen(path, "wb");
    if (!fh) {
        err = std::string("cannot open ") + path;
        return false;
    }
    std::string out;
    out.reserve(1 << 20);
    out += format == 0 ? kTsvHeader : kPinHeader;
    bool ok = true;
    for (uint64_t r = 0; r < n && ok; ++r) {
        const uint64_t i = order ? order[r] : r;
        const SageFeature& x = f[i];
        if (x.peptide_idx >= db.n_peptides() || x.file_id >= n_files) {
            err = "feature " + std::to_string(i) + ": peptide index or file id out of range";
            ok = false;
            break;
        }
        const uint64_t pep = x.peptide_idx;
        const float rt = x.rt;
        const float aligned_rt = col(pc.aligned_rt, i, rt);  // Feature defaults: scoring.rs:576-592
        const float predicted_rt = col(pc.predicted_rt, i, 0.0f), delta_rt = col(pc.delta_rt_model, i, 0.999f);
        const float predicted_ims = col(pc.predicted_ims, i, 0.0f), delta_ims = col(pc.delta_ims_model, i, 0.999f);
        const float posterior_error = col(pc.posterior_error, i, 1.0f);
        if (format == 0) {  // serialize_feature, runner.rs:687-780
            itoa(out, psm_id[i]);
            str(out, db.peptide_string(pep));
            str(out, db.peptide_proteins(pep));
            out += '\t';  // protein_groups: None
            itoa(out, db.pep_protein_off[pep + 1] - db.pep_protein_off[pep]);
            itoa(out, 0);  // num_protein_groups
            str(out, filenames[x.file_id]);
            str(out, spec_ids[i]);
            itoa(out, x.rank);
            itoa(out, x.label);
            f32(out, x.expmass);
            f32(out, x.calcmass);
            itoa(out, (unsigned)x.charge);
            itoa(out, x.peptide_len);
            itoa(out, (unsigned)x.missed_cleavages);
            itoa(out, (unsigned)db.semi[pep]);
            f32(out, x.isotope_error);
            f32(out, x.delta_mass);
            f32(out, x.average_ppm);
            f64(out, x.hyperscore);
            f64(out, x.delta_next);
            f64(out, x.delta_best);
            f32(out, rt);
            f32(out, aligned_rt);
            f32(out, predicted_rt);
            f32(out, delta_rt);
            f32(out, x.ims);
            f32(out, predicted_ims);
            f32(out, delta_ims);
            itoa(out, x.matched_peaks);
            itoa(out, x.longest_b);
            itoa(out, x.longest_y);
            f32(out, x.longest_y_pct);
            f32(out, x.matched_intensity_pct);
            itoa(out, x.scored_candidates);
            f64(out, x.poisson);
            f32(out, col(pc.discriminant_score, i, 0.0f));
            f32(out, posterior_error);
            f32(out, col(pc.spectrum_q, i, 1.0f));
            f32(out, col(pc.peptide_q, i, 1.0f));
            f32(out, col(pc.protein_q, i, 1.0f));
            f32(out, 1.0f);  // protein_group_q
            f32(out, x.ms2_intensity);
        } else {  // serialize_pin, runner.rs:938-1084
            const unsigned z = x.charge;
            itoa(out, psm_id[i]);
            itoa(out, x.label);
            str(out, scan_number(spec_ids[i]));
            f32(out, x.expmass);
            f32(out, x.calcmass);
            str(out, filenames[x.file_id]);
            f32(out, rt);
            f32(out, x.ims);
            itoa(out, x.rank);
            for (unsigned c = 2; c <= 6; ++c) itoa(out, (int)(z == c));
            itoa(out, (z < 2 || z > 6) ? z : 0u);
            itoa(out, x.peptide_len);
            itoa(out, (unsigned)x.missed_cleavages);
            itoa(out, (unsigned)db.semi[pep]);
            f32(out, x.isotope_error);
            f32(out, std::log1p(std::fabs(x.delta_mass)));  // f32::ln_1p == log1pf
            f32(out, x.average_ppm);
            f64(out, std::log1p(x.hyperscore));
            f64(out, std::log1p(x.delta_next));
            f64(out, std::log1p(x.delta_best));
            f32(out, aligned_rt);
            f32(out, predicted_rt);
            const float clamped = delta_rt < 0.001f ? 0.001f : (delta_rt > 1.0f ? 1.0f : delta_rt);
            f32(out, std::sqrt(clamped));
            f32(out, predicted_ims);
            f32(out, delta_ims);
            itoa(out, x.matched_peaks);
            itoa(out, x.longest_b);
            itoa(out, x.longest_y);
            f32(out, x.longest_y_pct);
            f32(out, std::log1p(x.matched_intensity_pct));
            itoa(out, x.scored_candidates);
            f64(out, std::log1p(-x.poisson));
            f32(out, posterior_error);
            str(out, db.peptide_string(pep));
            str(out, db.peptide_proteins(pep));
        }
        out.back() = '\n';  // the last field's tab
        if (out.size() >= (1 << 20)) {
            ok = std::fwrite(out.data(), 1, out.size(), fh) == out.size();
            out.clear();
        }
    }
    if (ok && !out.empty()) ok = std::fwrite(out.data(), 1, out.size(), fh) == out.size();
    if (std::fclose(fh) != 0) ok = false;
    if (!ok && err.empty()) err = std::string("write to ") + path + " failed";
    return ok;
}

}  // namespace sagehip
