// Host-side IndexedDatabase construction (product).  See host_db.hpp.
// Reference: crates/sage/src/{fasta,enzyme,peptide,modification,ion_series,database}.rs
#include "host_db.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <numeric>
#include <stdexcept>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <unordered_set>

namespace sagehip {

unsigned host_threads() {
    if (const char* e = std::getenv("SAGE_HIP_THREADS")) {
        int v = std::atoi(e);
        if (v > 0) return (unsigned)v;
    }
    unsigned n = std::thread::hardware_concurrency();
    return n ? n : 1;
}

// SAGE_HIP_TIMING=1: wall time of the stages of the host-side database build on stderr
struct StageTimer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    const bool on = std::getenv("SAGE_HIP_TIMING") != nullptr;
    void lap(const char* what) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[host_db] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

// dynamic-chunk parallel loop over [0, n): f(begin, end, thread_index)
void parallel_for(size_t n, size_t grain, const std::function<void(size_t, size_t, unsigned)>& f) {
    const unsigned nt = (unsigned)std::min<size_t>(host_threads(), (n + grain - 1) / std::max<size_t>(grain, 1));
    if (nt <= 1) {
        if (n) f(0, n, 0);
        return;
    }
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; t++)
        pool.emplace_back([&, t]() {
            for (;;) {
                const size_t b = next.fetch_add(grain);
                if (b >= n) break;
                f(b, std::min(n, b + grain), t);
            }
        });
    for (auto& th : pool) th.join();
}

namespace {

// Stable sort on the host threads: equal pieces are sorted independently, then merged pairwise, level by level (the merges of
// a level run in parallel).  Same result as std::stable_sort — pieces and merges both keep the order of equal elements.
template <typename T, typename Less>
void parallel_stable_sort(std::vector<T>& v, Less less) {
    const size_t n = v.size();
    unsigned pieces = 1;
    while (pieces < host_threads() && n / (pieces * 2) >= 65536) pieces *= 2;
    if (pieces == 1) {
        std::stable_sort(v.begin(), v.end(), less);
        return;
    }
    std::vector<size_t> cut(pieces + 1);
    for (unsigned i = 0; i <= pieces; i++) cut[i] = n * i / pieces;
    parallel_for(pieces, 1, [&](size_t b, size_t e, unsigned) {
        for (size_t i = b; i < e; i++) std::stable_sort(v.begin() + cut[i], v.begin() + cut[i + 1], less);
    });
    std::vector<T> tmp(n);
    std::vector<T>* src = &v;
    std::vector<T>* dst = &tmp;
    for (unsigned width = 1; width < pieces; width *= 2) {
        const unsigned pairs = pieces / (2 * width);
        parallel_for(pairs, 1, [&](size_t b, size_t e, unsigned) {
            for (size_t k = b; k < e; k++) {
                const size_t lo = cut[2 * width * k], mid = cut[2 * width * k + width], hi = cut[2 * width * (k + 1)];
                std::merge(std::make_move_iterator(src->begin() + lo), std::make_move_iterator(src->begin() + mid),
                           std::make_move_iterator(src->begin() + mid), std::make_move_iterator(src->begin() + hi),
                           dst->begin() + lo, less);  // (std::merge takes from the first range on ties: stable)
            }
        });
        std::swap(src, dst);
    }
    if (src != &v) v.swap(tmp);
}

// destroys the elements of a vector of heap-owning objects on the host threads (millions of small frees)
template <typename T>
void parallel_clear(std::vector<T>& v) {
    parallel_for(v.size(), 16384, [&](size_t b, size_t e, unsigned) {
        for (size_t i = b; i < e; i++) {
            T gone = std::move(v[i]);
            (void)gone;
        }
    });
    std::vector<T>().swap(v);
}

constexpr float kH2O = 18.010565f;     // mass.rs:5
constexpr float kProton = 1.0072764f;  // mass.rs:6
constexpr float kNeutron = 1.00335f;   // mass.rs:7

const float kResidue[26] = {  // mass.rs:64-68
    71.03711f,  0.0f,       103.00919f, 115.02694f, 129.04259f, 147.0684f,  57.02146f,
    137.05891f, 113.08406f, 0.0f,       128.09496f, 113.08406f, 131.0405f,  114.04293f,
    237.14774f, 97.05276f,  128.05858f, 156.1011f,  87.03203f,  101.04768f, 150.95363f,
    99.06841f,  186.07932f, 0.0f,       163.06332f, 0.0f};

inline int32_t f32_order_key(float f) {  // f32::total_cmp as a signed-int compare
    int32_t i;
    std::memcpy(&i, &f, 4);
    return i ^ (int32_t)(((uint32_t)(i >> 31)) >> 1);
}

bool is_valid_aa(uint8_t c) {  // mass.rs:59-62
    return c != 0 && std::strchr("ACDEFGHIKLMNPQRSTVWYUO", c) != nullptr;
}

enum Pos : uint8_t { kNterm = 0, kCterm = 1, kFull = 2, kInternal = 3 };  // enzyme.rs:64-71

// One enzymatic digest product (enzyme.rs:13-26), referencing its protein by id.
struct Cut {
    uint32_t protein;
    uint32_t start, len;
    uint8_t missed;
    uint8_t pos;
    bool semi;
    bool decoy;
};

// A candidate peptide while the list is still being sorted/deduped (peptide.rs:12-31).
struct Pep {
    std::string seq;
    std::vector<float> mods;
    float nterm = NAN, cterm = NAN;  // NaN == None
    float mono = 0.f;
    uint8_t missed = 0;
    uint8_t pos = kInternal;
    bool decoy = false;
    bool semi = false;  // peptide.rs:28 semi_enzymatic
    std::vector<uint32_t> proteins;
};

struct SiteRef {  // peptide.rs:336-341
    uint8_t kind;  // 0 N-term, 1 C-term, 2 residue
    uint32_t index;
    float mass;
};

// ---- fasta.rs:16-56 ----------------------------------------------------------------------
void parse_fasta(const std::string& text, const std::string& decoy_tag, bool generate_decoys,
                 std::vector<std::string>& names, std::vector<std::string>& seqs) {
    std::string header, acc_seq;
    auto emit = [&]() {
        size_t a = 0;
        while (a < header.size() && isspace((unsigned char)header[a])) a++;
        size_t b = a;
        while (b < header.size() && !isspace((unsigned char)header[b])) b++;
        std::string acc = header.substr(a, b - a);
        if (acc.find(decoy_tag) == std::string::npos || !generate_decoys) {
            names.push_back(acc);
            seqs.push_back(acc_seq);
        }
        acc_seq.clear();
    };
    size_t i = 0, n = text.size();
    while (i < n) {
        size_t e = text.find('\n', i);
        if (e == std::string::npos) e = n;
        size_t le = e;
        if (le > i && text[le - 1] == '\r') le--;
        if (le > i) {  // non-empty line
            size_t a = i, b = le;
            while (a < b && isspace((unsigned char)text[a])) a++;
            while (b > a && isspace((unsigned char)text[b - 1])) b--;
            if (a < b && text[a] == '>') {
                if (!acc_seq.empty()) emit();
                header.assign(text, a + 1, b - a - 1);
            } else {
                acc_seq.append(text, a, b - a);
            }
        }
        i = e + 1;
    }
    if (!acc_seq.empty()) emit();
}

// ---- enzyme.rs:189-342 -------------------------------------------------------------------
struct EnzymeSpec {
    bool present = false;  // false => non-specific digest (enzyme == None)
    bool dollar = false;
    bool cleave[26] = {}, skip[26] = {};
    bool c_terminal = true, semi = false;
};

EnzymeSpec make_enzyme(const DbBuildConfig& cfg) {  // enzyme.rs:135-187
    EnzymeSpec e;
    if (cfg.cleave_at.empty()) return e;
    e.present = true;
    if (cfg.cleave_at == "$") {
        e.dollar = true;
        return e;
    }
    for (unsigned char c : cfg.cleave_at)
        if (c >= 'A' && c <= 'Z') e.cleave[c - 'A'] = true;
    for (unsigned char c : cfg.restrict_)
        if (c >= 'A' && c <= 'Z') e.skip[c - 'A'] = true;
    e.c_terminal = cfg.c_terminal;
    e.semi = cfg.semi_enzymatic;
    return e;
}

struct Span {
    uint32_t start, end;
    uint8_t missed;
    bool semi;
};

void digest_protein(const std::string& seq, uint32_t protein, bool protein_is_decoy, const EnzymeSpec& enz,
                    const DbBuildConfig& cfg, std::vector<Cut>& out) {
    const uint32_t n = (uint32_t)seq.size();
    std::vector<Span> spans;
    if (!enz.present) {  // enzyme.rs:224-238
        for (size_t len = cfg.min_len; len <= cfg.max_len; len++) {
            uint32_t last = n >= len ? n - (uint32_t)len : 0;
            for (uint32_t i = 0; i <= last; i++) spans.push_back({i, i + (uint32_t)len, 0, false});
        }
    } else {
        uint32_t left = 0;
        auto boundary = [&](uint32_t right) {
            if (right < n) {
                unsigned char nx = (unsigned char)seq[right];
                if (nx >= 'A' && nx <= 'Z' && enz.skip[nx - 'A']) return;
            }
            spans.push_back({left, right, 0, false});
            left = right;
        };
        if (enz.dollar) {
            boundary(n);
        } else {
            for (uint32_t i = 0; i < n; i++) {
                unsigned char c = (unsigned char)seq[i];
                if (c >= 'A' && c <= 'Z' && enz.cleave[c - 'A']) boundary(enz.c_terminal ? i + 1 : i);
            }
        }
        spans.push_back({left, n, 0, false});
        const size_t base = spans.size();
        if (cfg.missed_cleavages > 0) {  // enzyme.rs:242-257
            for (uint32_t w = 1; w <= 1u + cfg.missed_cleavages; w++)
                for (size_t s = 0; s + w <= base; s++)
                    spans.push_back({spans[s].start, spans[s + w - 1].end, (uint8_t)(w - 1), false});
        }
        if (enz.semi) {  // enzyme.rs:266-287
            const size_t upto = spans.size();
            for (size_t s = 0; s < upto; s++) {
                Span sp = spans[s];
                for (uint32_t cut = sp.start; cut < sp.end; cut++) {
                    spans.push_back({sp.start, cut, sp.missed, true});
                    spans.push_back({cut, sp.end, sp.missed, true});
                }
            }
        }
    }
    std::unordered_set<std::string_view> seen;
    for (const Span& sp : spans) {
        if (sp.start > sp.end || sp.end > n) continue;
        uint32_t len = sp.end - sp.start;
        if (len < cfg.min_len || len > cfg.max_len || len == 0) continue;
        std::string_view sv(seq.data() + sp.start, len);
        if (!seen.insert(sv).second) continue;
        uint8_t pos = (sp.start == 0 && sp.end == n) ? kFull : sp.start == 0 ? kNterm : sp.end == n ? kCterm : kInternal;
        out.push_back({protein, sp.start, len, sp.missed, pos, sp.semi, protein_is_decoy});
    }
}

// ---- peptide.rs:136-305 ------------------------------------------------------------------
template <class F>
void visit_sites(const Pep& p, ModTarget t, F&& f) {  // push_resi / static_mods site selection
    const uint8_t first = p.seq.empty() ? 0 : (uint8_t)p.seq.front();
    const uint8_t last = p.seq.empty() ? 0 : (uint8_t)p.seq.back();
    const uint32_t last_ix = p.seq.empty() ? 0 : (uint32_t)p.seq.size() - 1;
    const bool prot_n = p.pos == kNterm || p.pos == kFull, prot_c = p.pos == kCterm || p.pos == kFull;
    switch (t.where) {
        case ModTarget::PeptideN:
        case ModTarget::ProteinN:
            if (t.where == ModTarget::ProteinN && !prot_n) return;
            if (t.residue < 0) f((uint8_t)0, 0u);
            else if ((uint8_t)t.residue == first) f((uint8_t)2, 0u);
            return;
        case ModTarget::PeptideC:
        case ModTarget::ProteinC:
            if (t.where == ModTarget::ProteinC && !prot_c) return;
            if (t.residue < 0) f((uint8_t)1, 0u);
            else if ((uint8_t)t.residue == last) f((uint8_t)2, last_ix);
            return;
        case ModTarget::Residue:
            for (uint32_t i = 0; i < p.seq.size(); i++)
                if ((uint8_t)p.seq[i] == (uint8_t)t.residue) f((uint8_t)2, i);
            return;
    }
}

inline void put_site(Pep& p, uint8_t kind, uint32_t index, float mass) {  // apply_site, peptide.rs:136-153
    if (kind == 0) {
        if (std::isnan(p.nterm)) p.nterm = 0.0f + mass;
    } else if (kind == 1) {
        if (std::isnan(p.cterm)) p.cterm = 0.0f + mass;
    } else if (p.mods[index] == 0.0f) {
        p.mods[index] += mass;
    }
}

void finish_mods(Pep& p, const DbBuildConfig& cfg) {  // static mods + mass, peptide.rs:266-269 / 295-301
    for (const auto& sm : cfg.static_mods) {
        if (sm.first.where == ModTarget::Residue) {  // peptide.rs:248-254: `= mass`, not `+=`
            for (size_t i = 0; i < p.seq.size(); i++)
                if ((uint8_t)p.seq[i] == (uint8_t)sm.first.residue && p.mods[i] == 0.0f) p.mods[i] = sm.second;
        } else {
            visit_sites(p, sm.first, [&](uint8_t k, uint32_t ix) { put_site(p, k, ix, sm.second); });
        }
    }
    float msum = 0.0f;  // modification_mass, peptide.rs:129-133
    for (float m : p.mods) msum += m;
    msum = msum + (std::isnan(p.nterm) ? 0.0f : p.nterm) + (std::isnan(p.cterm) ? 0.0f : p.cterm);
    p.mono += msum;
}

void expand_mods(const Pep& base, const DbBuildConfig& cfg, std::vector<Pep>& out) {  // Peptide::apply
    const size_t first = out.size();
    out.push_back(base);
    if (!cfg.variable_mods.empty()) {
        std::vector<SiteRef> sites;
        for (const auto& vm : cfg.variable_mods)
            visit_sites(base, vm.first, [&](uint8_t k, uint32_t ix) { sites.push_back({k, ix, vm.second}); });
        const size_t m = sites.size();
        std::vector<size_t> pick;
        for (size_t n = 1; n <= cfg.max_variable_mods && n <= m; n++) {
            pick.resize(n);
            std::iota(pick.begin(), pick.end(), 0);
            for (;;) {
                // no_duplicates (peptide.rs:321-333) + unique-site check (peptide.rs:279-284)
                bool ok = true;
                for (size_t a = 0; a < n && ok; a++)
                    for (size_t b = a + 1; b < n; b++) {
                        const SiteRef &x = sites[pick[a]], &y = sites[pick[b]];
                        if (x.kind == y.kind && (x.kind != 2 || x.index == y.index)) { ok = false; break; }
                    }
                if (ok) {
                    Pep q = base;
                    for (size_t a = 0; a < n; a++) put_site(q, sites[pick[a]].kind, sites[pick[a]].index, sites[pick[a]].mass);
                    out.push_back(std::move(q));
                }
                size_t i = n;  // advance to the next lexicographic combination
                while (i > 0 && pick[i - 1] == m - n + (i - 1)) i--;
                if (i == 0) break;
                pick[i - 1]++;
                for (size_t j = i; j < n; j++) pick[j] = pick[j - 1] + 1;
            }
        }
    }
    for (size_t i = first; i < out.size(); i++) finish_mods(out[i], cfg);
}

inline int cmp_f32_partial(float a, float b) { return a < b ? -1 : (a > b ? 1 : 0); }
inline int cmp_opt_nan(float a, float b) {  // Option<f32>::partial_cmp, None(NaN) < Some
    const bool na = std::isnan(a), nb = std::isnan(b);
    if (na || nb) return na == nb ? 0 : (na ? -1 : 1);
    return cmp_f32_partial(a, b);
}

// database.rs:226-230: monoisotopic.total_cmp, then Peptide::initial_sort (peptide.rs:34-52)
bool pep_before(const Pep& a, const Pep& b) {
    int32_t ka = f32_order_key(a.mono), kb = f32_order_key(b.mono);
    if (ka != kb) return ka < kb;
    int c = a.seq.compare(b.seq);
    if (c != 0) return c < 0;
    size_t n = std::min(a.mods.size(), b.mods.size());
    bool unordered = false;
    for (size_t i = 0; i < n; i++) {
        if (a.mods[i] < b.mods[i]) return true;
        if (a.mods[i] > b.mods[i]) return false;
        if (!(a.mods[i] == b.mods[i])) { unordered = true; break; }
    }
    if (!unordered && a.mods.size() != b.mods.size()) return a.mods.size() < b.mods.size();
    c = cmp_opt_nan(a.nterm, b.nterm);
    if (c != 0) return c < 0;
    return cmp_opt_nan(a.cterm, b.cterm) < 0;
}

inline bool opt_eq(float a, float b) {
    const bool na = std::isnan(a), nb = std::isnan(b);
    return (na && nb) || (!na && !nb && a == b);
}

// LSD radix sort of (key, value) pairs by 32-bit key, stable.
void radix_sort_pairs(std::vector<uint32_t>& keys, std::vector<uint32_t>& vals) {
    const size_t n = keys.size();
    std::vector<uint32_t> k2(n), v2(n);
    for (int pass = 0; pass < 4; pass++) {
        const int shift = pass * 8;
        size_t hist[256] = {};
        for (size_t i = 0; i < n; i++) hist[(keys[i] >> shift) & 255]++;
        size_t sum = 0;
        for (int b = 0; b < 256; b++) { size_t c = hist[b]; hist[b] = sum; sum += c; }
        for (size_t i = 0; i < n; i++) {
            size_t d = hist[(keys[i] >> shift) & 255]++;
            k2[d] = keys[i];
            v2[d] = vals[i];
        }
        keys.swap(k2);
        vals.swap(v2);
    }
}

}  // namespace

float residue_mass(uint8_t aa) { return (aa >= 'A' && aa <= 'Z') ? kResidue[aa - 'A'] : 0.0f; }

bool parse_mod_target(const char* key, ModTarget& out) {  // modification.rs:66-104
    const size_t len = std::strlen(key);
    if (len == 0 || len > 2) return false;
    auto terminal = [&](ModTarget::Where w) {
        out.where = w;
        out.residue = len > 1 ? (int)(uint8_t)key[1] : -1;
        return true;
    };
    switch (key[0]) {
        case '^': return terminal(ModTarget::PeptideN);
        case '$': return terminal(ModTarget::PeptideC);
        case '[': return terminal(ModTarget::ProteinN);
        case ']': return terminal(ModTarget::ProteinC);
        default: break;
    }
    if (!is_valid_aa((uint8_t)key[0])) return false;
    out.where = ModTarget::Residue;
    out.residue = (uint8_t)key[0];
    return true;
}

DbBuildConfig config_from_params(const SageDbParams& p) {  // Builder::make_parameters, database.rs:96-115
    DbBuildConfig c;
    uint64_t b = p.bucket_size ? p.bucket_size : 8192, pow2 = 1;
    while (pow2 < b) pow2 <<= 1;
    c.bucket_size = pow2;
    if (p.enzyme_present) {  // From<EnzymeBuilder>, database.rs:43-57
        c.missed_cleavages = p.missed_cleavages >= 0 ? (uint8_t)p.missed_cleavages : 1;
        c.min_len = p.min_len >= 0 ? (size_t)p.min_len : 5;
        c.max_len = p.max_len >= 0 ? (size_t)p.max_len : 50;
        c.cleave_at = p.cleave_at ? p.cleave_at : "KR";
        c.restrict_ = p.restrict_ ? p.restrict_ : "";
        c.c_terminal = p.c_terminal >= 0 ? p.c_terminal != 0 : true;
        c.semi_enzymatic = p.semi_enzymatic >= 0 ? p.semi_enzymatic != 0 : false;
    }  // else EnzymeBuilder::default(), database.rs:29-41 (the DbBuildConfig defaults)
    c.peptide_min_mass = p.peptide_min_mass;
    c.peptide_max_mass = p.peptide_max_mass;
    if (p.ion_kinds && p.n_ion_kinds) c.ion_kinds.assign(p.ion_kinds, p.ion_kinds + p.n_ion_kinds);
    c.min_ion_index = p.min_ion_index;
    for (uint32_t i = 0; i < p.n_static_mods; i++) {
        ModTarget t;
        if (!parse_mod_target(p.static_mod_keys[i], t)) continue;
        bool replaced = false;
        for (auto& sm : c.static_mods)
            if (sm.first.where == t.where && sm.first.residue == t.residue) { sm.second = p.static_mod_masses[i]; replaced = true; }
        if (!replaced) c.static_mods.push_back({t, p.static_mod_masses[i]});
    }
    // The reference iterates a HashMap here (random order, database.rs:178-182 / peptide.rs:265);
    // a fixed order keeps the build reproducible: by (kind, residue).
    std::sort(c.static_mods.begin(), c.static_mods.end(), [](const auto& a, const auto& b) {
        return a.first.where != b.first.where ? a.first.where < b.first.where : a.first.residue < b.first.residue;
    });
    // variable mods: group masses by target in first-appearance order, then flatten
    std::vector<std::pair<ModTarget, std::vector<float>>> grouped;
    for (uint32_t i = 0; i < p.n_var_mods; i++) {
        ModTarget t;
        if (!parse_mod_target(p.var_mod_keys[i], t)) continue;
        bool found = false;
        for (auto& g : grouped)
            if (g.first.where == t.where && g.first.residue == t.residue) { g.second.push_back(p.var_mod_masses[i]); found = true; }
        if (!found) grouped.push_back({t, {p.var_mod_masses[i]}});
    }
    for (auto& g : grouped)
        for (float m : g.second) c.variable_mods.push_back({g.first, m});
    c.max_variable_mods = std::max<uint64_t>(p.max_variable_mods, 1);
    c.decoy_tag = p.decoy_tag ? p.decoy_tag : "rev_";
    c.generate_decoys = p.generate_decoys != 0;
    c.peptides_only = p.peptides_only != 0;
    return c;
}

void ion_series_flat(const uint8_t* seq, const float* mods, size_t len, float nterm, float mono, uint8_t kind,
                     float* out) {  // ion_series.rs:36-85
    const float C = 12.0f, O = 15.994914f, H = 1.007825f, PRO = 1.0072764f, N = 14.003074f;
    const float NH3 = N + H * 2.0f + PRO;
    const float nt = std::isnan(nterm) ? 0.0f : nterm;
    float cum;
    switch (kind) {
        case SAGE_ION_A: cum = nt - (C + O); break;
        case SAGE_ION_B: cum = nt; break;
        case SAGE_ION_C: cum = nt + NH3; break;
        case SAGE_ION_X: cum = mono - nt + (C + O - NH3 + N + H); break;
        case SAGE_ION_Y: cum = mono - nt; break;
        default: cum = mono - nt - NH3; break;
    }
    const bool forward = kind <= SAGE_ION_C;
    for (size_t i = 0; i + 1 < len; i++) {
        const float step = residue_mass(seq[i]) + mods[i];
        cum += forward ? step : -step;
        out[i] = cum;
    }
}

namespace {

void init_database(HostDb& db, const DbBuildConfig& cfg) {
    db.bucket_size = cfg.bucket_size;
    db.ion_kinds = cfg.ion_kinds;
    db.decoy_tag = cfg.decoy_tag;
    db.generate_decoys = cfg.generate_decoys;
    db.min_ion_index = cfg.min_ion_index;
}

// Fasta::digest (fasta.rs:58-79) for db.protein_names / prot_seqs: one Cut per enzymatic product
std::vector<Cut> digest_fasta(const HostDb& db, const std::vector<std::string>& prot_seqs, const DbBuildConfig& cfg) {
    const EnzymeSpec enz = make_enzyme(cfg);
    const size_t n_prot = prot_seqs.size();
    std::vector<std::vector<Cut>> per_protein(n_prot);
    parallel_for(n_prot, 16, [&](size_t b, size_t e, unsigned) {
        for (size_t i = b; i < e; i++) {
            const bool tagged = db.protein_names[i].find(cfg.decoy_tag) != std::string::npos;
            if (tagged && cfg.generate_decoys) continue;  // fasta.rs:66-72
            digest_protein(prot_seqs[i], (uint32_t)i, tagged, enz, cfg, per_protein[i]);
        }
    });
    std::vector<Cut> cuts;
    for (auto& v : per_protein) cuts.insert(cuts.end(), v.begin(), v.end());
    return cuts;
}

void finish_database(HostDb& db, std::vector<Pep>& peps, const DbBuildConfig& cfg);

}  // namespace

// Parameters::build (database.rs:260-263) over targets [first_target, first_target + n_targets) of the FASTA: the whole
// file by default, one chunk of Fasta::iter_chunks (fasta.rs:81-89) for the prefilter flow
HostDb build_database(const std::string& fasta_text, const DbBuildConfig& cfg, uint64_t first_target, uint64_t n_targets) {
    HostDb db;
    StageTimer timer;
    init_database(db, cfg);

    std::vector<std::string> prot_seqs;
    parse_fasta(fasta_text, cfg.decoy_tag, cfg.generate_decoys, db.protein_names, prot_seqs);
    timer.lap("parse_fasta");
    if (first_target || n_targets < prot_seqs.size()) {
        const size_t lo = std::min<size_t>(first_target, prot_seqs.size());
        const size_t hi = std::min<size_t>(prot_seqs.size(), lo + std::min<uint64_t>(n_targets, prot_seqs.size()));
        db.protein_names = std::vector<std::string>(db.protein_names.begin() + lo, db.protein_names.begin() + hi);
        prot_seqs = std::vector<std::string>(prot_seqs.begin() + lo, prot_seqs.begin() + hi);
    }

    // 1. digest every protein (fasta.rs:58-79), proteins in parallel
    std::vector<Cut> cuts = digest_fasta(db, prot_seqs, cfg);
    timer.lap("digest");
    auto cut_seq = [&](const Cut& c) { return std::string_view(prot_seqs[c.protein].data() + c.start, c.len); };

    // 2. group_digests (enzyme.rs:33-62): by (position, decoy, sequence)
    parallel_stable_sort(cuts, [&](const Cut& a, const Cut& b) {
        if (a.pos != b.pos) return a.pos < b.pos;
        if (a.decoy != b.decoy) return a.decoy < b.decoy;
        return cut_seq(a) < cut_seq(b);
    });
    std::vector<size_t> group_start;
    for (size_t i = 0; i < cuts.size(); i++) {
        if (i == 0 || cuts[i].pos != cuts[i - 1].pos || cuts[i].decoy != cuts[i - 1].decoy ||
            cut_seq(cuts[i]) != cut_seq(cuts[i - 1]))
            group_start.push_back(i);
    }
    group_start.push_back(cuts.size());
    const size_t n_groups = group_start.size() - 1;
    timer.lap("group_digests (sort)");

    // target sequences, for dropping decoys that collide with a target (database.rs:184-190, 212)
    std::unordered_set<std::string_view> target_seqs;
    target_seqs.reserve(n_groups);
    for (size_t g = 0; g < n_groups; g++)
        if (!cuts[group_start[g]].decoy) target_seqs.insert(cut_seq(cuts[group_start[g]]));

    timer.lap("target sequence set");
    // 3. modify + decoys (database.rs:193-214), groups in parallel
    const unsigned nthreads = host_threads();
    std::vector<std::vector<Pep>> per_thread(nthreads);
    parallel_for(n_groups, 256, [&](size_t gb, size_t ge, unsigned tid) {
        std::vector<Pep>& mine = per_thread[tid];
        std::vector<Pep> forms;
        for (size_t g = gb; g < ge; g++) {
            const Cut& ref = cuts[group_start[g]];
            std::string_view sv = cut_seq(ref);
            Pep base;  // TryFrom<Digest>, peptide.rs:357-388
            float mass = kH2O;
            bool valid = true;
            for (unsigned char ch : sv) {
                if (ch >= 0x80) { valid = false; break; }
            }
            if (valid)
                for (unsigned char ch : sv) {
                    float r = residue_mass(ch);
                    if (r == 0.0f) { valid = false; break; }
                    mass += r;
                }
            if (!valid) continue;
            base.seq.assign(sv);
            base.mods.assign(sv.size(), 0.0f);
            base.mono = mass;
            base.missed = ref.missed;
            base.pos = ref.pos;
            base.decoy = ref.decoy;
            base.semi = ref.semi;
            for (size_t i = group_start[g]; i < group_start[g + 1]; i++) base.proteins.push_back(cuts[i].protein);
            forms.clear();
            expand_mods(base, cfg, forms);
            for (Pep& f : forms) {
                if (!(f.mono >= cfg.peptide_min_mass && f.mono <= cfg.peptide_max_mass)) continue;
                if (cfg.generate_decoys) {  // Peptide::reverse, peptide.rs:307-318
                    Pep r = f;
                    r.decoy = !f.decoy;
                    const size_t n = r.seq.empty() ? 0 : r.seq.size() - 1;
                    if (n > 1) {
                        std::reverse(r.seq.begin() + 1, r.seq.begin() + n);
                        std::reverse(r.mods.begin() + 1, r.mods.begin() + n);
                    }
                    if (!r.decoy || target_seqs.find(std::string_view(r.seq)) == target_seqs.end())
                        mine.push_back(std::move(r));
                }
                if (!f.decoy || target_seqs.find(std::string_view(f.seq)) == target_seqs.end())
                    mine.push_back(std::move(f));
            }
        }
    });
    std::vector<Pep> peps;
    {
        size_t total = 0;
        for (auto& v : per_thread) total += v.size();
        peps.reserve(total);
        for (auto& v : per_thread) {
            for (auto& p : v) peps.push_back(std::move(p));
            std::vector<Pep>().swap(v);
        }
    }

    timer.lap("modify + decoys");
    finish_database(db, peps, cfg);
    timer.lap("finish_database");
    return db;
}

namespace {

// reorder_peptides (database.rs:221-258) then Parameters::build_from_peptides (database.rs:265-346) into the flat layout
void finish_database(HostDb& db, std::vector<Pep>& peps, const DbBuildConfig& cfg) {
    StageTimer timer;
    // 4. reorder_peptides (database.rs:221-258): sort, dedup, merge proteins.  The comparator is a
    // total order on the dedup key, so the result does not depend on the (thread-dependent) input order
    // except for which duplicate's missed_cleavages/position survives; keep that deterministic by
    // preferring the smallest (missed, position, first protein) among duplicates.
    parallel_stable_sort(peps, [](const Pep& a, const Pep& b) {
        if (pep_before(a, b)) return true;
        if (pep_before(b, a)) return false;
        if (a.missed != b.missed) return a.missed < b.missed;
        if (a.pos != b.pos) return a.pos < b.pos;
        return a.proteins < b.proteins;
    });
    timer.lap("  reorder: sort");
    std::vector<Pep> uniq;
    uniq.reserve(peps.size());
    for (Pep& p : peps) {
        if (!uniq.empty()) {
            Pep& k = uniq.back();
            if (p.mono == k.mono && p.seq == k.seq && p.mods == k.mods && opt_eq(p.nterm, k.nterm) &&
                opt_eq(p.cterm, k.cterm)) {
                k.proteins.insert(k.proteins.end(), p.proteins.begin(), p.proteins.end());
                k.decoy = k.decoy && p.decoy;
                continue;
            }
        }
        uniq.push_back(std::move(p));
    }
    parallel_clear(peps);
    timer.lap("  reorder: dedup");
    const size_t np = uniq.size();
    if (np >= 0xFFFFFFFFull) throw std::runtime_error("too many peptides for a u32 PeptideIx");

    // 5. flatten peptides
    db.pep_mono.resize(np);
    db.nterm.resize(np);
    db.cterm.resize(np);
    db.decoy.resize(np);
    db.missed.resize(np);
    db.semi.resize(np);
    db.position.resize(np);
    db.seq_off.resize(np + 1);
    db.pep_protein_off.resize(np + 1);
    uint64_t off = 0, poff = 0;
    for (size_t i = 0; i < np; i++) {
        db.seq_off[i] = off;
        off += uniq[i].seq.size();
        db.pep_protein_off[i] = poff;
        poff += uniq[i].proteins.size();
    }
    db.seq_off[np] = off;
    db.pep_protein_off[np] = poff;
    db.seq.resize(off);
    db.mods.resize(off);
    db.pep_protein_ids.resize(poff);
    parallel_for(np, 4096, [&](size_t ib, size_t ie, unsigned) {
      for (size_t i = ib; i < ie; i++) {
        Pep& p = uniq[i];
        db.pep_mono[i] = p.mono;
        db.nterm[i] = p.nterm;
        db.cterm[i] = p.cterm;
        db.decoy[i] = p.decoy;
        db.missed[i] = p.missed;
        db.semi[i] = p.semi;
        db.position[i] = p.pos;
        std::memcpy(db.seq.data() + db.seq_off[i], p.seq.data(), p.seq.size());
        std::memcpy(db.mods.data() + db.seq_off[i], p.mods.data(), p.mods.size() * 4);
        // proteins.sort_unstable() on names (database.rs:248-250); ids are in FASTA order, so sort by name
        std::sort(p.proteins.begin(), p.proteins.end(), [&](uint32_t a, uint32_t b) {
            return db.protein_names[a] < db.protein_names[b];
        });
        std::copy(p.proteins.begin(), p.proteins.end(), db.pep_protein_ids.begin() + db.pep_protein_off[i]);
      }
    });
    timer.lap("  flatten");
    parallel_clear(uniq);
    timer.lap("  free");
    if (cfg.peptides_only) return;  // the device builds the fragment index (index_build.hip)

    // 6. theoretical fragments (database.rs:272-297)
    const size_t nk = cfg.ion_kinds.size();
    std::vector<uint64_t> frag_off(np + 1, 0);
    for (size_t i = 0; i < np; i++) {
        const size_t len = db.seq_off[i + 1] - db.seq_off[i];
        const size_t lm1 = len ? len - 1 : 0;
        // kept ions per kind: (idx+1) > min  |  (lm1 - idx) > min, idx in [0, lm1)
        const size_t kept = lm1 > cfg.min_ion_index ? lm1 - cfg.min_ion_index : 0;
        frag_off[i + 1] = frag_off[i] + kept * nk;
    }
    const uint64_t nf = frag_off[np];
    std::vector<uint32_t> keys(nf), peps_of(nf);
    parallel_for(np, 1024, [&](size_t ib, size_t ie, unsigned) {
        std::vector<float> ions;
        for (size_t i = ib; i < ie; i++) {
            const size_t len = db.seq_off[i + 1] - db.seq_off[i];
            const size_t lm1 = len ? len - 1 : 0;
            ions.resize(lm1);
            uint64_t w = frag_off[i];
            for (size_t k = 0; k < nk; k++) {
                const uint8_t kind = cfg.ion_kinds[k];
                ion_series_flat(db.seq.data() + db.seq_off[i], db.mods.data() + db.seq_off[i], len, db.nterm[i],
                                db.pep_mono[i], kind, ions.data());
                for (size_t idx = 0; idx < lm1; idx++) {
                    const bool keep = kind <= SAGE_ION_C ? (idx + 1) > cfg.min_ion_index : (lm1 - idx) > cfg.min_ion_index;
                    if (!keep) continue;
                    keys[w] = (uint32_t)f32_order_key(ions[idx]) ^ 0x80000000u;  // unsigned radix order
                    peps_of[w] = (uint32_t)i;
                    w++;
                }
            }
        }
    });
    // 7. global m/z sort (database.rs:301); equal m/z keep ascending peptide order (stable)
    radix_sort_pairs(keys, peps_of);
    db.fragments.resize(nf);
    const uint64_t nb = (nf + cfg.bucket_size - 1) / cfg.bucket_size;
    db.min_value.resize(nb);
    auto key_to_f32 = [](uint32_t k) {
        int32_t i = (int32_t)(k ^ 0x80000000u);
        i ^= (int32_t)(((uint32_t)(i >> 31)) >> 1);
        float f;
        std::memcpy(&f, &i, 4);
        return f;
    };
    // 8. per bucket: record the minimum, then order by peptide (database.rs:337-346)
    parallel_for(nb, 8, [&](size_t bb, size_t be, unsigned) {
        std::vector<std::pair<uint32_t, uint32_t>> tmp;
        for (size_t b = bb; b < be; b++) {
            const uint64_t s = (uint64_t)b * cfg.bucket_size, e = std::min<uint64_t>(s + cfg.bucket_size, nf);
            db.min_value[b] = key_to_f32(keys[s]);
            tmp.resize(e - s);
            for (uint64_t i = s; i < e; i++) tmp[i - s] = {peps_of[i], keys[i]};
            std::sort(tmp.begin(), tmp.end());  // (peptide, m/z key): equals a stable sort by peptide
            for (uint64_t i = s; i < e; i++) db.fragments[i] = {tmp[i - s].first, key_to_f32(tmp[i - s].second)};
        }
    });
}

}  // namespace

// Fasta::parse(..).targets.len() (fasta.rs:16-56)
uint64_t fasta_num_targets(const std::string& fasta_text, const DbBuildConfig& cfg) {
    std::vector<std::string> names, seqs;
    parse_fasta(fasta_text, cfg.decoy_tag, cfg.generate_decoys, names, seqs);
    return names.size();
}

// Parameters::auto_calculate_prefilter_chunk_size (database.rs:142-160)
uint64_t prefilter_chunk_size(const std::string& fasta_text, const DbBuildConfig& cfg, uint64_t requested) {
    if (requested) return requested;
    HostDb db;
    std::vector<std::string> prot_seqs;
    parse_fasta(fasta_text, cfg.decoy_tag, cfg.generate_decoys, db.protein_names, prot_seqs);
    const uint64_t total_unmodified = digest_fasta(db, prot_seqs, cfg).size();
    std::vector<std::pair<uint8_t, int>> targets;  // variable_mods.len(): distinct ModificationSpecificity keys
    for (auto& m : cfg.variable_mods) {
        std::pair<uint8_t, int> key{(uint8_t)m.first.where, m.first.residue};
        if (std::find(targets.begin(), targets.end(), key) == targets.end()) targets.push_back(key);
    }
    const uint64_t mod_count_estimate = (targets.size() + 1) * (1ull << cfg.max_variable_mods);
    const uint64_t chunk_count = mod_count_estimate * total_unmodified / (1ull << 23);
    return chunk_count == 0 ? prot_seqs.size() : prot_seqs.size() / chunk_count;
}

// runner.rs:215-238: the peptides every chunk database kept (keep[c][ix] != 0), concatenated, through reorder_peptides —
// which merges a peptide found in several chunks, joins its proteins and clears `decoy` when any copy is a target
// (database.rs:233-247) — and build_from_peptides.  Chunks are consecutive slices of the FASTA targets, so a chunk-local
// protein id becomes global by adding the number of proteins of the chunks before it.
HostDb merge_kept(const std::vector<const HostDb*>& chunks, const std::vector<const uint8_t*>& keep, const DbBuildConfig& cfg) {
    HostDb db;
    init_database(db, cfg);
    std::vector<Pep> peps;
    for (size_t c = 0; c < chunks.size(); c++) {
        const HostDb& src = *chunks[c];
        const uint32_t base = (uint32_t)db.protein_names.size();
        db.protein_names.insert(db.protein_names.end(), src.protein_names.begin(), src.protein_names.end());
        for (uint64_t i = 0; i < src.n_peptides(); i++) {
            if (!keep[c][i]) continue;
            Pep p;
            p.seq.assign((const char*)src.seq.data() + src.seq_off[i], src.seq_off[i + 1] - src.seq_off[i]);
            p.mods.assign(src.mods.begin() + src.seq_off[i], src.mods.begin() + src.seq_off[i + 1]);
            p.nterm = src.nterm[i];
            p.cterm = src.cterm[i];
            p.mono = src.pep_mono[i];
            p.missed = src.missed[i];
            p.pos = src.position[i];
            p.decoy = src.decoy[i] != 0;
            p.semi = src.semi[i] != 0;
            for (uint64_t j = src.pep_protein_off[i]; j < src.pep_protein_off[i + 1]; j++)
                p.proteins.push_back(base + src.pep_protein_ids[j]);
            peps.push_back(std::move(p));
        }
    }
    finish_database(db, peps, cfg);
    return db;
}

SageDbView HostDb::view() const {
    SageDbView v{};
    v.fragments = fragments.data();
    v.n_fragments = fragments.size();
    v.min_value = min_value.data();
    v.n_buckets = min_value.size();
    v.bucket_size = bucket_size;
    v.pep_mono = pep_mono.data();
    v.seq_off = seq_off.data();
    v.seq = seq.data();
    v.mods = mods.data();
    v.nterm = nterm.data();
    v.cterm = cterm.data();
    v.decoy = decoy.data();
    v.missed_cleavages = missed.data();
    v.n_peptides = pep_mono.size();
    v.ion_kinds = ion_kinds.data();
    v.n_ion_kinds = (uint32_t)ion_kinds.size();
    v.min_ion_index = min_ion_index;
    if (fragments.empty()) v.fragments = nullptr;
    return v;
}

static std::string signed_mass(float m) {  // "{:+}" of an f32 (peptide.rs:393-405)
    char buf[96];
    for (int dec = 0; dec <= 12; dec++) {
        std::snprintf(buf, sizeof buf, "%+.*f", dec, (double)m);
        if (std::strtof(buf, nullptr) == m) break;
    }
    return buf;
}

std::string HostDb::peptide_string(uint64_t i) const {
    std::string s;
    if (!std::isnan(nterm[i])) s += "[" + signed_mass(nterm[i]) + "]-";
    for (uint64_t j = seq_off[i]; j < seq_off[i + 1]; j++) {
        s += (char)seq[j];
        if (mods[j] != 0.0f) s += "[" + signed_mass(mods[j]) + "]";
    }
    if (!std::isnan(cterm[i])) s += "-[" + signed_mass(cterm[i]) + "]";
    return s;
}

// The map keys of the picked competitions (fdr.rs:123-187) as dense ids in order of first appearance.
// peptide: `peptide.to_string()`, of `peptide.reverse()` for generated decoys (fdr.rs:126-132; reverse() flips
// sequence[1..len-1] and the residue modifications with it, peptide.rs:307-318) — a target and the decoy derived from it
// share a key.  protein: the `proteins` vector of peptides with exactly one protein (fdr.rs:158-161), i.e. the protein id.
void HostDb::competition_keys(const uint32_t* peptide_idx, uint64_t n, uint32_t* peptide_key, uint32_t& n_peptide_keys,
                              uint32_t* protein_key, uint32_t& n_protein_keys) const {
    std::unordered_map<std::string, uint32_t> pep_ids;
    // protein competitions are keyed by the `proteins` vector's CONTENT (fdr.rs:158-161), i.e. the accession string: two FASTA
    // entries with one accession are one competition
    std::unordered_map<std::string, uint32_t> prot_ids;
    std::unordered_map<uint32_t, uint32_t> prot_seen;  // protein id -> key
    std::unordered_map<uint32_t, uint32_t> seen;  // peptide index -> key (PSMs of one peptide are common)
    for (uint64_t f = 0; f < n; f++) {
        const uint64_t i = peptide_idx[f];
        auto hit = seen.find((uint32_t)i);
        if (hit != seen.end()) {
            peptide_key[f] = hit->second;
        } else {
            std::string s;
            const uint64_t lo = seq_off[i], len = seq_off[i + 1] - lo, last = len ? len - 1 : 0;
            const bool flip = generate_decoys && decoy[i] && last > 1;
            if (!std::isnan(nterm[i])) s += "[" + signed_mass(nterm[i]) + "]-";
            for (uint64_t j = 0; j < len; j++) {
                const uint64_t src = lo + ((flip && j >= 1 && j < last) ? last - j : j);  // s[1..last].reverse()
                s += (char)seq[src];
                if (mods[src] != 0.0f) s += "[" + signed_mass(mods[src]) + "]";
            }
            if (!std::isnan(cterm[i])) s += "-[" + signed_mass(cterm[i]) + "]";
            const uint32_t id = pep_ids.emplace(std::move(s), (uint32_t)pep_ids.size()).first->second;
            seen.emplace((uint32_t)i, id);
            peptide_key[f] = id;
        }
        if (pep_protein_off[i + 1] - pep_protein_off[i] == 1) {
            const uint32_t prot = pep_protein_ids[pep_protein_off[i]];
            auto ph = prot_seen.find(prot);
            if (ph == prot_seen.end()) {
                const uint32_t id = prot_ids.emplace(protein_names[prot], (uint32_t)prot_ids.size()).first->second;
                ph = prot_seen.emplace(prot, id).first;
            }
            protein_key[f] = ph->second;
        } else {
            protein_key[f] = 0xFFFFFFFFu;
        }
    }
    n_peptide_keys = (uint32_t)pep_ids.size();
    n_protein_keys = (uint32_t)prot_ids.size();
}

std::string HostDb::peptide_proteins(uint64_t i) const {  // Peptide::proteins, peptide.rs:81-97
    std::string s;
    for (uint64_t j = pep_protein_off[i]; j < pep_protein_off[i + 1]; j++) {
        if (j > pep_protein_off[i]) s += ';';
        if (decoy[i] && generate_decoys) s += decoy_tag;
        s += protein_names[pep_protein_ids[j]];
    }
    return s;
}

// ---- spectrum.rs:179-227, 279-412 --------------------------------------------------------------
uint64_t process_ms2(uint64_t take_top_n, bool deisotope, float min_deisotope_mz, const float* mz,
                     const float* intensity, uint64_t n, uint8_t precursor_charge, float* out_mass,
                     float* out_intensity, float* out_tic) {
    struct Pk {
        float inten, mass;
    };
    std::vector<Pk> kept;
    if (deisotope) {
        const uint8_t max_charge = precursor_charge ? precursor_charge : 3;  // spectrum.rs:289-293
        std::vector<float> acc(intensity, intensity + n);  // envelope-summed intensity
        std::vector<uint8_t> charge(n, 0);                 // 0 == None
        std::vector<uint8_t> child(n, 0);                  // 1 == has an envelope parent
        const float ppm = 10.0f;
        for (uint64_t i = n; i-- > 0;) {  // spectrum.rs:198-225
            uint64_t j = i ? i - 1 : 0;
            while (mz[i] - mz[j] <= kNeutron + ppm * mz[i] / 1000000.0f && mz[j] >= min_deisotope_mz) {
                const float delta = mz[i] - mz[j];
                const float tol = ppm * mz[i] / 1000000.0f;
                for (unsigned z = 1; z <= max_charge; z++) {
                    const float iso = kNeutron / (float)z;
                    if (std::fabs(delta - iso) <= tol && intensity[i] < intensity[j]) {
                        if (charge[i] && charge[i] != z) continue;
                        acc[j] += acc[i];
                        charge[j] = (uint8_t)z;
                        charge[i] = (uint8_t)z;
                        child[i] = 1;
                    }
                }
                j = j ? j - 1 : 0;
                if (j == 0) break;
            }
        }
        std::vector<uint32_t> order(n);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {  // spectrum.rs:303-307
            int32_t ia = f32_order_key(acc[a]), ib = f32_order_key(acc[b]);
            if (ia != ib) return ia > ib;
            return f32_order_key(mz[a]) < f32_order_key(mz[b]);
        });
        for (uint32_t ix : order) {
            if (child[ix]) continue;
            if (kept.size() >= take_top_n) break;
            kept.push_back({acc[ix], (mz[ix] - kProton) * (float)(charge[ix] ? charge[ix] : 1)});
        }
    } else {
        kept.resize(n);
        for (uint64_t i = 0; i < n; i++) kept[i] = {intensity[i], (mz[i] - kProton) * 1.0f};
        // bounded_min_heapify + truncate (spectrum.rs:332-333, heap.rs:7-28); the later mass sort is
        // stable, so the heap layout is observable only through equal-mass peaks: keep it exact.
        auto less = [](const Pk& a, const Pk& b) {
            int32_t ia = f32_order_key(a.inten), ib = f32_order_key(b.inten);
            if (ia != ib) return ia < ib;
            return f32_order_key(a.mass) < f32_order_key(b.mass);
        };
        const size_t k = take_top_n;
        if (kept.size() > k) {
            auto sift = [&](size_t idx) {
                for (;;) {
                    size_t l = 2 * idx + 1, r = l + 1, s = idx;
                    if (l < k && less(kept[l], kept[s])) s = l;
                    if (r < k && less(kept[r], kept[s])) s = r;
                    if (s == idx) break;
                    std::swap(kept[s], kept[idx]);
                    idx = s;
                }
            };
            for (size_t i = k / 2; i-- > 0;) sift(i);
            for (size_t i = k; i < kept.size(); i++)
                if (less(kept[0], kept[i])) { std::swap(kept[0], kept[i]); sift(0); }
            kept.resize(k);
        }
    }
    std::stable_sort(kept.begin(), kept.end(),
                     [](const Pk& a, const Pk& b) { return f32_order_key(a.mass) < f32_order_key(b.mass); });
    float tic = 0.0f;
    for (size_t i = 0; i < kept.size(); i++) {
        out_mass[i] = kept[i].mass;
        out_intensity[i] = kept[i].inten;
        tic += kept[i].inten;
    }
    *out_tic = tic;
    return kept.size();
}

}  // namespace sagehip
