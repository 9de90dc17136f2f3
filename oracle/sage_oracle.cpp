// ORACLE — TEST INFRASTRUCTURE ONLY (see sage_oracle.hpp header).
// Sequential restatement of the reference's search-and-score path.  Citations are
// /root/reference/crates/sage/src/<file>:<line>.
#include "sage_oracle.hpp"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <quadmath.h>
#include <set>
#include <unordered_set>

namespace sage_oracle {

// ---------------------------------------------------------------------------
// total_cmp — core::f32::total_cmp
// ---------------------------------------------------------------------------
int total_cmp(float a, float b) {
    int32_t l, r;
    std::memcpy(&l, &a, 4);
    std::memcpy(&r, &b, 4);
    l ^= (int32_t)(((uint32_t)(l >> 31)) >> 1);
    r ^= (int32_t)(((uint32_t)(r >> 31)) >> 1);
    return l < r ? -1 : (l > r ? 1 : 0);
}
int total_cmp(double a, double b) {
    int64_t l, r;
    std::memcpy(&l, &a, 8);
    std::memcpy(&r, &b, 8);
    l ^= (int64_t)(((uint64_t)(l >> 63)) >> 1);
    r ^= (int64_t)(((uint64_t)(r >> 63)) >> 1);
    return l < r ? -1 : (l > r ? 1 : 0);
}

// ---------------------------------------------------------------------------
// mass.rs
// ---------------------------------------------------------------------------
std::pair<float, float> Tolerance::bounds(float center) const {  // mass.rs:21-35
    switch (kind) {
        case PPM: {
            float delta_lo = center * lo / 1000000.0f;
            float delta_hi = center * hi / 1000000.0f;
            return {center + delta_lo, center + delta_hi};
        }
        case PCT: {
            float delta_lo = center * lo / 100.0f;
            float delta_hi = center * hi / 100.0f;
            return {center + delta_lo, center + delta_hi};
        }
        default:
            return {center + lo, center + hi};
    }
}
bool Tolerance::contains(float center, float rhs) const {
    auto b = bounds(center);
    return rhs >= b.first && rhs <= b.second;
}
Tolerance Tolerance::scaled(float rhs) const { return {kind, lo * rhs, hi * rhs}; }  // mass.rs:47-57

static const float MONOISOTOPIC_MASSES[26] = {  // mass.rs:64-68
    71.03711f,  0.0f,       103.00919f, 115.02694f, 129.04259f, 147.0684f,  57.02146f,
    137.05891f, 113.08406f, 0.0f,       128.09496f, 113.08406f, 131.0405f,  114.04293f,
    237.14774f, 97.05276f,  128.05858f, 156.1011f,  87.03203f,  101.04768f, 150.95363f,
    99.06841f,  186.07932f, 0.0f,       163.06332f, 0.0f};

float monoisotopic(uint8_t aa) {  // mass.rs:70-76
    if (aa >= 'A' && aa <= 'Z') return MONOISOTOPIC_MASSES[aa - 'A'];
    return 0.0f;
}
bool valid_aa(uint8_t aa) {  // mass.rs:59-62
    static const char* V = "ACDEFGHIKLMNPQRSTVWYUO";
    return aa != 0 && std::strchr(V, (int)aa) != nullptr;
}

// ---------------------------------------------------------------------------
// modification.rs:66-104
// ---------------------------------------------------------------------------
bool parse_modspec(const std::string& s, ModSpec& out) {
    if (s.size() > 2) return false;  // TooLong
    auto term = [&](ModSpec::Type t) {
        out.type = t;
        out.residue = s.size() > 1 ? (int)(uint8_t)s[1] : -1;
        return true;
    };
    if (!s.empty() && s[0] == '^') return term(ModSpec::PeptideN);
    if (!s.empty() && s[0] == '$') return term(ModSpec::PeptideC);
    if (!s.empty() && s[0] == '[') return term(ModSpec::ProteinN);
    if (!s.empty() && s[0] == ']') return term(ModSpec::ProteinC);
    if (s.empty()) return false;  // Empty
    if (valid_aa((uint8_t)s[0])) {
        out.type = ModSpec::Residue;
        out.residue = (uint8_t)s[0];
        return true;
    }
    return false;  // InvalidResidue
}

// ---------------------------------------------------------------------------
// enzyme.rs
// ---------------------------------------------------------------------------
std::optional<Enzyme> Enzyme::make(const std::string& cleave, const std::string& skip_suffix,
                                   bool c_terminal, bool semi_enzymatic) {  // enzyme.rs:135-187
    if (cleave.empty()) return std::nullopt;
    Enzyme e;
    if (cleave == "$") {
        e.dollar = true;
        e.c_terminal = true;
        e.semi_enzymatic = false;
        return e;
    }
    for (unsigned char c : cleave)
        if (c >= 'A' && c <= 'Z') e.cleave[c - 'A'] = true;
    for (unsigned char c : skip_suffix)
        if (c >= 'A' && c <= 'Z') e.skip_suffix[c - 'A'] = true;
    e.c_terminal = c_terminal;
    e.semi_enzymatic = semi_enzymatic;
    return e;
}

std::vector<DigestSite> Enzyme::cleavage_sites(const std::string& sequence) const {  // enzyme.rs:189-217
    std::vector<DigestSite> sites;
    size_t left = 0;
    auto on_match = [&](size_t mstart, size_t mend) {
        size_t right = c_terminal ? mend : mstart;
        if (right < sequence.size()) {
            unsigned char b = (unsigned char)sequence[right];
            if (b >= 'A' && b <= 'Z' && skip_suffix[b - 'A']) return;
        }
        sites.push_back({left, right, 0, false});
        left = right;
    };
    if (dollar) {
        on_match(sequence.size(), sequence.size());  // regex "$": one empty match at end of text
    } else {
        for (size_t i = 0; i < sequence.size(); i++) {
            unsigned char c = (unsigned char)sequence[i];
            if (c >= 'A' && c <= 'Z' && cleave[c - 'A']) on_match(i, i + 1);
        }
    }
    sites.push_back({left, sequence.size(), 0, false});
    return sites;
}

std::vector<DigestSite> EnzymeParameters::cleavage_sites(const std::string& sequence) const {  // :221-240
    if (enzyme) return enzyme->cleavage_sites(sequence);
    std::vector<DigestSite> v;
    for (size_t len = min_len; len <= max_len; len++) {
        size_t last = sequence.size() >= len ? sequence.size() - len : 0;  // saturating_sub
        for (size_t i = 0; i <= last; i++) v.push_back({i, i + len, 0, false});
    }
    return v;
}

std::vector<Digest> EnzymeParameters::digest(const std::string& sequence,
                                             const std::string& protein) const {  // enzyme.rs:289-342
    size_t n = sequence.size();
    std::vector<Digest> digests;
    std::vector<DigestSite> sites = cleavage_sites(sequence);
    uint8_t mc = enzyme ? missed_cleavages : 0;

    if (mc > 0) {  // missed_cleavage_sites, enzyme.rs:242-257
        std::vector<DigestSite> extra;
        for (unsigned cleavage = 1; cleavage <= 1u + mc; cleavage++) {
            if (sites.size() < cleavage) continue;
            for (size_t w = 0; w + cleavage <= sites.size(); w++) {
                extra.push_back({sites[w].start, sites[w + cleavage - 1].end, (uint8_t)(cleavage - 1), false});
            }
        }
        sites.insert(sites.end(), extra.begin(), extra.end());
    }
    if (enzyme && enzyme->semi_enzymatic) {  // semi_enzymatic_sites, enzyme.rs:266-287
        std::vector<DigestSite> extra;
        for (const auto& site : sites) {
            for (size_t cut = site.start; cut < site.end; cut++) {
                extra.push_back({site.start, cut, site.missed_cleavages, true});
                extra.push_back({cut, site.end, site.missed_cleavages, true});
            }
        }
        sites.insert(sites.end(), extra.begin(), extra.end());
    }

    std::unordered_set<std::string> seen;
    for (const auto& site : sites) {
        if (site.start > site.end || site.end > n) continue;  // sequence.get(start..end) == None
        std::string sub = sequence.substr(site.start, site.end - site.start);
        size_t len = sub.size();
        Position position = (site.start == 0 && site.end == n)   ? Position::Full
                            : (site.start == 0)                  ? Position::Nterm
                            : (site.end == n)                    ? Position::Cterm
                                                                 : Position::Internal;
        if (len >= min_len && len <= max_len && len > 0 && seen.insert(sub).second) {
            Digest d;
            d.sequence = sub;
            d.missed_cleavages = site.missed_cleavages;
            d.decoy = false;
            d.semi_enzymatic = site.semi_enzymatic;
            d.position = position;
            d.protein = protein;
            digests.push_back(std::move(d));
        }
    }
    return digests;
}

std::vector<DigestGroup> group_digests(std::vector<Digest> digests) {  // enzyme.rs:33-62
    std::vector<DigestGroup> groups;
    if (digests.empty()) return groups;  // (the reference would panic on digests[0])
    // sort_unstable_by (position, decoy, sequence); ties are whole-key-equal => same group,
    // so a stable sort gives the same groups (the reference digest's other fields then come
    // from an arbitrary member; we take the first in input order)
    std::stable_sort(digests.begin(), digests.end(), [](const Digest& a, const Digest& b) {
        if (a.position != b.position) return a.position < b.position;
        if (a.decoy != b.decoy) return a.decoy < b.decoy;
        return a.sequence < b.sequence;
    });
    DigestGroup curr{digests[0], {}};
    for (auto& d : digests) {
        if (d.decoy == curr.reference.decoy && d.position == curr.reference.position &&
            d.sequence == curr.reference.sequence) {
            curr.proteins.push_back(d.protein);
        } else {
            std::sort(curr.proteins.begin(), curr.proteins.end());
            groups.push_back(std::move(curr));
            curr = DigestGroup{d, {d.protein}};
        }
    }
    groups.push_back(std::move(curr));
    return groups;
}

// ---------------------------------------------------------------------------
// fasta.rs
// ---------------------------------------------------------------------------
static std::string trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && std::isspace((unsigned char)s[a])) a++;
    while (b > a && std::isspace((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}
static std::string first_token(const std::string& s) {
    size_t a = 0;
    while (a < s.size() && std::isspace((unsigned char)s[a])) a++;
    size_t b = a;
    while (b < s.size() && !std::isspace((unsigned char)s[b])) b++;
    return s.substr(a, b - a);
}

Fasta Fasta::parse(const std::string& contents, const std::string& decoy_tag, bool generate_decoys) {
    Fasta f;  // fasta.rs:16-56
    f.decoy_tag = decoy_tag;
    f.generate_decoys = generate_decoys;
    std::string last_id, s;
    size_t pos = 0;
    auto flush = [&]() {
        std::string acc = first_token(last_id);
        if (acc.find(decoy_tag) == std::string::npos || !generate_decoys)
            f.targets.emplace_back(acc, s);
        s.clear();
    };
    while (pos <= contents.size()) {
        size_t nl = contents.find('\n', pos);
        std::string line = contents.substr(pos, nl == std::string::npos ? std::string::npos : nl - pos);
        if (!line.empty() && line.back() == '\r') line.pop_back();  // str::lines strips "\r\n"
        bool last = nl == std::string::npos;
        pos = last ? contents.size() + 1 : nl + 1;
        if (line.empty()) continue;
        line = trim(line);
        if (!line.empty() && line[0] == '>') {
            if (!s.empty()) flush();
            last_id = line.substr(1);
        } else {
            s += line;
        }
    }
    if (!s.empty()) flush();
    return f;
}

std::vector<Digest> Fasta::digest(const EnzymeParameters& enzyme) const {  // fasta.rs:58-79
    std::vector<Digest> out;
    for (const auto& t : targets) {
        bool is_decoy_protein = t.first.find(decoy_tag) != std::string::npos;
        for (auto& d : enzyme.digest(t.second, t.first)) {
            if (is_decoy_protein) {
                if (!generate_decoys) {
                    d.decoy = true;
                    out.push_back(std::move(d));
                }
            } else {
                out.push_back(std::move(d));
            }
        }
    }
    return out;
}

// ---------------------------------------------------------------------------
// peptide.rs
// ---------------------------------------------------------------------------
static int cmp_opt(const std::optional<float>& a, const std::optional<float>& b) {
    // Option<f32>::partial_cmp(...).unwrap_or(Equal): None < Some
    if (!a && !b) return 0;
    if (!a) return -1;
    if (!b) return 1;
    if (*a < *b) return -1;
    if (*a > *b) return 1;
    return 0;  // equal, or unordered -> Equal
}

int Peptide::initial_sort(const Peptide& o) const {  // peptide.rs:34-52
    int c = sequence.compare(o.sequence);
    if (c != 0) return c < 0 ? -1 : 1;
    // Vec<f32>::partial_cmp: lexicographic; first unordered pair -> None -> Equal
    size_t n = std::min(modifications.size(), o.modifications.size());
    for (size_t i = 0; i < n; i++) {
        float a = modifications[i], b = o.modifications[i];
        if (a < b) return -1;
        if (a > b) return 1;
        if (!(a == b)) { c = 0; goto after_mods; }  // NaN -> None -> unwrap_or(Equal)
    }
    if (modifications.size() != o.modifications.size())
        return modifications.size() < o.modifications.size() ? -1 : 1;
after_mods:
    c = cmp_opt(nterm, o.nterm);
    if (c != 0) return c;
    return cmp_opt(cterm, o.cterm);
}

bool Peptide::from_digest(const Digest& d, Peptide& out) {  // peptide.rs:357-388
    float mass = H2O;
    for (unsigned char c : d.sequence) {
        if (c >= 0x80) return false;
    }
    for (unsigned char c : d.sequence) {
        float mono = sage_oracle::monoisotopic(c);
        if (mono == 0.0f) return false;
        mass += mono;
    }
    out = Peptide{};
    out.decoy = d.decoy;
    out.position = d.position;
    out.modifications.assign(d.sequence.size(), 0.0f);
    out.sequence = d.sequence;
    out.monoisotopic = mass;
    out.missed_cleavages = d.missed_cleavages;
    out.semi_enzymatic = d.semi_enzymatic;
    out.proteins = {d.protein};
    return true;
}

namespace {
struct Site {  // peptide.rs:336-341: Nterm < Cterm < Sequence(u32)
    int kind;  // 0 Nterm, 1 Cterm, 2 Sequence
    uint32_t index;
    bool operator==(const Site& o) const { return kind == o.kind && index == o.index; }
};

void apply_site(Peptide& p, Site site, float mass) {  // peptide.rs:136-153
    switch (site.kind) {
        case 0:
            if (!p.nterm) p.nterm = 0.0f + mass;
            break;
        case 1:
            if (!p.cterm) p.cterm = 0.0f + mass;
            break;
        default:
            if (p.modifications[site.index] == 0.0f) p.modifications[site.index] += mass;
    }
}

// the shared (target, position) match of push_resi / static_mods: peptide.rs:155-206 / 208-256
// calls f(site) for every site selected by `target` on peptide `p`
template <class F>
void for_each_site(const Peptide& p, ModSpec target, F f) {
    uint8_t first = p.sequence.empty() ? 0 : (uint8_t)p.sequence.front();
    uint8_t last = p.sequence.empty() ? 0 : (uint8_t)p.sequence.back();
    uint32_t last_ix = (uint32_t)(p.sequence.empty() ? 0 : p.sequence.size() - 1);
    bool at_n = p.position == Position::Nterm || p.position == Position::Full;
    bool at_c = p.position == Position::Cterm || p.position == Position::Full;
    switch (target.type) {
        case ModSpec::PeptideN:
            if (target.residue < 0) f(Site{0, 0});
            else if ((uint8_t)target.residue == first) f(Site{2, 0});
            break;
        case ModSpec::PeptideC:
            if (target.residue < 0) f(Site{1, 0});
            else if ((uint8_t)target.residue == last) f(Site{2, last_ix});
            break;
        case ModSpec::ProteinN:
            if (!at_n) break;
            if (target.residue < 0) f(Site{0, 0});
            else if ((uint8_t)target.residue == first) f(Site{2, 0});
            break;
        case ModSpec::ProteinC:
            if (!at_c) break;
            if (target.residue < 0) f(Site{1, 0});
            else if ((uint8_t)target.residue == last) f(Site{2, last_ix});
            break;
        case ModSpec::Residue:
            for (size_t i = 0; i < p.sequence.size(); i++)
                if ((uint8_t)p.sequence[i] == (uint8_t)target.residue) f(Site{2, (uint32_t)i});
            break;
    }
}

void apply_static(Peptide& p, ModSpec target, float mass) {  // peptide.rs:208-256
    if (target.type == ModSpec::Residue) {
        for (size_t i = 0; i < p.sequence.size(); i++)
            if ((uint8_t)p.sequence[i] == (uint8_t)target.residue && p.modifications[i] == 0.0f)
                p.modifications[i] = mass;
        return;
    }
    for_each_site(p, target, [&](Site s) { apply_site(p, s, mass); });
}

float modification_mass(const Peptide& p) {  // peptide.rs:129-133
    float sum = 0.0f;
    for (float m : p.modifications) sum += m;
    return sum + p.nterm.value_or(0.0f) + p.cterm.value_or(0.0f);
}
}  // namespace

std::vector<Peptide> Peptide::apply(const std::vector<std::pair<ModSpec, float>>& variable_mods,
                                    const std::vector<std::pair<ModSpec, float>>& static_mods,
                                    size_t combinations) const {  // peptide.rs:258-305
    std::vector<Peptide> modified;
    if (variable_mods.empty()) {
        Peptide self = *this;
        for (const auto& sm : static_mods) apply_static(self, sm.first, sm.second);
        self.monoisotopic += modification_mass(self);
        modified.push_back(std::move(self));
        return modified;
    }
    std::vector<std::pair<Site, float>> mods;
    for (const auto& vm : variable_mods)
        for_each_site(*this, vm.first, [&](Site s) { mods.emplace_back(s, vm.second); });

    modified.push_back(*this);
    // itertools::combinations(n): lexicographic by index
    for (size_t n = 1; n <= combinations; n++) {
        if (n > mods.size()) break;
        std::vector<size_t> idx(n);
        for (size_t i = 0; i < n; i++) idx[i] = i;
        while (true) {
            // filter(no_duplicates): peptide.rs:321-333
            int nn = 0, cc = 0;
            for (size_t i : idx) {
                if (mods[i].first.kind == 0) nn++;
                if (mods[i].first.kind == 1) cc++;
            }
            bool ok = nn <= 1 && cc <= 1;
            if (ok) {  // site set must be unique (continue 'next)
                for (size_t a = 0; a < n && ok; a++)
                    for (size_t b = a + 1; b < n; b++)
                        if (mods[idx[a]].first == mods[idx[b]].first) { ok = false; break; }
            }
            if (ok) {
                Peptide pep = *this;
                for (size_t i : idx) apply_site(pep, mods[i].first, mods[i].second);
                modified.push_back(std::move(pep));
            }
            // next combination
            size_t i = n;
            while (i > 0 && idx[i - 1] == mods.size() - n + (i - 1)) i--;
            if (i == 0) break;
            idx[i - 1]++;
            for (size_t j = i; j < n; j++) idx[j] = idx[j - 1] + 1;
        }
    }
    for (auto& pep : modified) {
        for (const auto& sm : static_mods) apply_static(pep, sm.first, sm.second);
        pep.monoisotopic += modification_mass(pep);
    }
    return modified;
}

Peptide Peptide::reverse() const {  // peptide.rs:307-318
    Peptide pep = *this;
    pep.decoy = !decoy;
    size_t n = pep.sequence.empty() ? 0 : pep.sequence.size() - 1;
    if (n > 1) {
        std::reverse(pep.sequence.begin() + 1, pep.sequence.begin() + n);
        std::reverse(pep.modifications.begin() + 1, pep.modifications.begin() + n);
    }
    return pep;
}

static std::string fmt_plus(float m) {  // "{:+}" for f32: shortest round-trip decimal, explicit sign
    char buf[96];
    for (int dec = 0; dec <= 12; dec++) {
        std::snprintf(buf, sizeof buf, "%.*f", dec, (double)m);
        if (std::strtof(buf, nullptr) == m) break;
    }
    std::string s = buf;
    if (s[0] != '-' && s[0] != '+') s = "+" + s;
    return s;
}

std::string Peptide::to_string() const {  // peptide.rs:391-408
    std::string out;
    if (nterm) out += "[" + fmt_plus(*nterm) + "]-";
    for (size_t i = 0; i < sequence.size(); i++) {
        out += sequence[i];
        if (modifications[i] != 0.0f) out += "[" + fmt_plus(modifications[i]) + "]";
    }
    if (cterm) out += "-[" + fmt_plus(*cterm) + "]";
    return out;
}

// ---------------------------------------------------------------------------
// ion_series.rs:36-85
// ---------------------------------------------------------------------------
void ion_series_into(const Peptide& p, Kind kind, std::vector<float>& out);
std::vector<float> ion_series(const Peptide& p, Kind kind) {
    std::vector<float> out;
    ion_series_into(p, kind, out);
    return out;
}
void ion_series_into(const Peptide& p, Kind kind, std::vector<float>& out) {
    out.clear();
    const float C = 12.0f, O = 15.994914f, H = 1.007825f, PRO = 1.0072764f, N = 14.003074f;
    const float NH3 = N + H * 2.0f + PRO;
    float nterm = p.nterm.value_or(0.0f);
    float cum;
    switch (kind) {
        case Kind::A: cum = nterm - (C + O); break;
        case Kind::B: cum = nterm; break;
        case Kind::C: cum = nterm + NH3; break;
        case Kind::X: cum = p.monoisotopic - nterm + (C + O - NH3 + N + H); break;
        case Kind::Y: cum = p.monoisotopic - nterm; break;
        default: cum = p.monoisotopic - nterm - NH3; break;  // Z
    }
    if (p.sequence.empty()) return;
    for (size_t idx = 0; idx + 1 < p.sequence.size(); idx++) {
        float r = monoisotopic((uint8_t)p.sequence[idx]);
        float m = p.modifications[idx];
        if (is_nterm_kind(kind)) cum += r + m; else cum += -(r + m);
        out.push_back(cum);
    }
}

// ---------------------------------------------------------------------------
// database.rs
// ---------------------------------------------------------------------------
EnzymeBuilder EnzymeBuilder::defaults() {  // database.rs:29-41
    EnzymeBuilder e;
    e.missed_cleavages = 0;
    e.min_len = 5;
    e.max_len = 50;
    e.cleave_at = "KR";
    e.restrict = "P";
    e.c_terminal = true;
    e.semi_enzymatic = false;
    return e;
}
EnzymeParameters EnzymeBuilder::to_parameters() const {  // database.rs:43-57
    EnzymeParameters p;
    p.missed_cleavages = missed_cleavages.value_or(1);
    p.min_len = min_len.value_or(5);
    p.max_len = max_len.value_or(50);
    p.enzyme = Enzyme::make(cleave_at.value_or("KR"), restrict.value_or(""), c_terminal.value_or(true),
                            semi_enzymatic.value_or(false));
    return p;
}

std::vector<Peptide> Parameters::digest(const Fasta& fasta) const {  // database.rs:162-219
    EnzymeParameters enz = enzyme.to_parameters();
    std::vector<Digest> digests = fasta.digest(enz);
    std::vector<DigestGroup> groups = group_digests(std::move(digests));

    std::vector<std::pair<ModSpec, float>> mods;
    for (const auto& vm : variable_mods)
        for (float m : vm.second) mods.emplace_back(vm.first, m);

    std::unordered_set<std::string> targets;
    for (const auto& g : groups)
        if (!g.reference.decoy) targets.insert(g.reference.sequence);

    std::vector<Peptide> target_decoys;
    for (const auto& g : groups) {
        Peptide base;
        if (!Peptide::from_digest(g.reference, base)) continue;
        base.proteins = g.proteins;  // TryFrom<DigestGroup>, peptide.rs:347-355
        for (auto& pep : base.apply(mods, static_mods, max_variable_mods)) {
            if (!(pep.monoisotopic >= peptide_min_mass && pep.monoisotopic <= peptide_max_mass)) continue;
            std::vector<Peptide> pair;
            if (generate_decoys) pair.push_back(pep.reverse());
            pair.push_back(std::move(pep));
            for (auto& q : pair) {
                if (!q.decoy || targets.find(q.sequence) == targets.end())
                    target_decoys.push_back(std::move(q));
            }
        }
    }
    reorder_peptides(target_decoys);
    return target_decoys;
}

void Parameters::reorder_peptides(std::vector<Peptide>& v) {  // database.rs:221-258
    std::stable_sort(v.begin(), v.end(), [](const Peptide& a, const Peptide& b) {
        int c = total_cmp(a.monoisotopic, b.monoisotopic);
        if (c != 0) return c < 0;
        return a.initial_sort(b) < 0;
    });
    // Vec::dedup_by(|remove, keep| ...)
    std::vector<Peptide> out;
    for (auto& remove : v) {
        if (!out.empty()) {
            Peptide& keep = out.back();
            if (remove.monoisotopic == keep.monoisotopic && remove.sequence == keep.sequence &&
                remove.modifications == keep.modifications && remove.nterm == keep.nterm &&
                remove.cterm == keep.cterm) {
                keep.proteins.insert(keep.proteins.end(), remove.proteins.begin(), remove.proteins.end());
                keep.decoy = keep.decoy && remove.decoy;
                continue;
            }
        }
        out.push_back(std::move(remove));
    }
    for (auto& p : out) std::sort(p.proteins.begin(), p.proteins.end());
    v = std::move(out);
}

IndexedDatabase Parameters::build(const Fasta& fasta) const {  // database.rs:260-263
    return build_from_peptides(digest(fasta));
}

IndexedDatabase Parameters::build_from_peptides(std::vector<Peptide> target_decoys) const {  // :265-364
    std::vector<Theoretical> fragments;
    for (size_t idx = 0; idx < target_decoys.size(); idx++) {
        const Peptide& peptide = target_decoys[idx];
        size_t lm1 = peptide.sequence.empty() ? 0 : peptide.sequence.size() - 1;
        for (Kind kind : ion_kinds) {
            std::vector<float> ions = ion_series(peptide, kind);
            for (size_t ion_idx = 0; ion_idx < ions.size(); ion_idx++) {
                bool keep = is_nterm_kind(kind) ? (ion_idx + 1) > min_ion_index
                                                : (lm1 - ion_idx) > min_ion_index;
                if (keep) fragments.push_back({(uint32_t)idx, ions[ion_idx]});
            }
        }
    }
    // par_sort_unstable_by(fragment_mz.total_cmp): ties are broken arbitrarily in the
    // reference; peptide_index as secondary key makes this restatement deterministic
    std::sort(fragments.begin(), fragments.end(), [](const Theoretical& a, const Theoretical& b) {
        int c = total_cmp(a.fragment_mz, b.fragment_mz);
        if (c != 0) return c < 0;
        return a.peptide_index < b.peptide_index;
    });
    std::vector<float> min_value;
    for (size_t s = 0; s < fragments.size(); s += bucket_size) {
        size_t e = std::min(s + bucket_size, fragments.size());
        min_value.push_back(fragments[s].fragment_mz);
        std::stable_sort(fragments.begin() + s, fragments.begin() + e,
                         [](const Theoretical& a, const Theoretical& b) {
                             return a.peptide_index < b.peptide_index;
                         });
    }
    IndexedDatabase db;
    db.peptides = std::move(target_decoys);
    db.fragments = std::move(fragments);
    db.min_value = std::move(min_value);
    db.bucket_size = bucket_size;
    db.ion_kinds = ion_kinds;
    db.generate_decoys = generate_decoys;
    db.decoy_tag = decoy_tag;
    return db;
}

IndexedQuery IndexedDatabase::query(float precursor_mass, Tolerance precursor_tol,
                                    Tolerance fragment_tol) const {  // database.rs:402-425
    auto b = precursor_tol.bounds(precursor_mass);
    auto r = binary_search_slice(
        peptides.data(), peptides.size(),
        [](const Peptide& p, const float& bound) { return total_cmp(p.monoisotopic, bound); }, b.first,
        b.second);
    return IndexedQuery{this, precursor_mass, precursor_tol, fragment_tol, r.first, r.second};
}

void WorkCounters::add(const WorkCounters& o) {
    queries += o.queries; page_searches += o.page_searches; pages += o.pages; scanned += o.scanned;
    hits += o.hits; peaks += o.peaks; rescored += o.rescored; rescored_residues += o.rescored_residues;
    reported += o.reported; algorithmic_bytes += o.algorithmic_bytes;
}

static uint64_t ceil_log2(uint64_t x) {  // ceil(log2(x)) for x >= 1
    uint64_t n = 0;
    while ((1ull << n) < x) n++;
    return n;
}

// database.rs:480-536, with the consumer of the yielded fragments as a template parameter: the scoring loop passes its
// closure directly (inlined, like the reference's iterator chain); IndexedQuery::page_search wraps it for other callers
template <class F>
static inline void page_search_impl(const IndexedQuery& q, float mass, F&& f, WorkCounters* wc) {
    const IndexedDatabase* db = q.db;
    const Tolerance& fragment_tol = q.fragment_tol;
    const Tolerance& precursor_tol = q.precursor_tol;
    const float precursor_mass = q.precursor_mass;
    const size_t pre_idx_lo = q.pre_idx_lo, pre_idx_hi = q.pre_idx_hi;
    auto fb = fragment_tol.bounds(mass);
    auto pb = precursor_tol.bounds(precursor_mass);
    float fragment_lo = fb.first, fragment_hi = fb.second;
    float precursor_lo = pb.first, precursor_hi = pb.second;

    auto pages = binary_search_slice(
        db->min_value.data(), db->min_value.size(),
        [](const float& m, const float& bound) { return total_cmp(m, bound); }, fragment_lo, fragment_hi);
    if (wc) {
        wc->page_searches++;
        wc->algorithmic_bytes += 2 * ceil_log2(db->min_value.size() + 1) * 4;
    }
    for (size_t page = pages.first; page < pages.second; page++) {
        size_t left_idx = page * db->bucket_size;
        size_t right_idx = std::min((page + 1) * db->bucket_size, db->fragments.size());
        const Theoretical* slice = db->fragments.data() + left_idx;
        size_t slen = right_idx - left_idx;
        auto inner = binary_search_slice(
            slice, slen,
            [](const Theoretical& fr, const size_t& bound) {
                size_t v = fr.peptide_index;
                return v < bound ? -1 : (v > bound ? 1 : 0);
            },
            pre_idx_lo, pre_idx_hi);
        if (wc) {
            wc->pages++;
            wc->scanned += inner.second - inner.first;
            wc->algorithmic_bytes += 2 * ceil_log2(db->bucket_size + 1) * 8 + 8 * (inner.second - inner.first);
        }
        for (size_t i = inner.first; i < inner.second; i++) {
            const Theoretical& frag = slice[i];
            bool ok = (frag.peptide_index > (uint32_t)pre_idx_lo ||
                       (frag.peptide_index == (uint32_t)pre_idx_lo &&
                        db->peptides[frag.peptide_index].monoisotopic >= precursor_lo)) &&
                      (frag.peptide_index < (uint32_t)pre_idx_hi ||
                       (frag.peptide_index == (uint32_t)pre_idx_hi &&
                        db->peptides[frag.peptide_index].monoisotopic <= precursor_hi)) &&
                      frag.fragment_mz >= fragment_lo && frag.fragment_mz <= fragment_hi;
            if (ok) {
                if (wc) wc->hits++;
                f(frag);
            }
        }
    }
}

void IndexedQuery::page_search(float mass, const std::function<void(const Theoretical&)>& f,
                               WorkCounters* wc) const {
    page_search_impl(*this, mass, f, wc);
}

// ---------------------------------------------------------------------------
// spectrum.rs
// ---------------------------------------------------------------------------
long select_most_intense_peak(const float* masses, const float* intensities, size_t n, float center,
                              Tolerance tolerance, std::optional<float> offset) {  // spectrum.rs:134-159
    auto b = tolerance.bounds(center);
    float lo = b.first + offset.value_or(0.0f);
    float hi = b.second + offset.value_or(0.0f);
    auto r = binary_search_slice(
        masses, n, [](const float& m, const float& q) { return total_cmp(m, q); }, lo, hi);
    long best = -1;
    float max_int = 0.0f;
    for (size_t idx = r.first; idx < r.second; idx++) {
        if (!(masses[idx] >= lo && masses[idx] <= hi)) continue;
        if (intensities[idx] >= max_int) {
            max_int = intensities[idx];
            best = (long)idx;
        }
    }
    return best;
}

std::vector<Deisotoped> deisotope(const float* mz, const float* inten, size_t n, uint8_t max_charge,
                                  float ppm, float min_mz) {  // spectrum.rs:179-227
    std::vector<Deisotoped> peaks(n);
    for (size_t i = 0; i < n; i++) peaks[i] = {mz[i], inten[i], std::nullopt, std::nullopt};
    for (size_t i = n; i-- > 0;) {
        size_t j = i == 0 ? 0 : i - 1;
        while (mz[i] - mz[j] <= NEUTRON + Tolerance::ppm_to_delta_mass(mz[i], ppm) && mz[j] >= min_mz) {
            float delta = mz[i] - mz[j];
            float tol = Tolerance::ppm_to_delta_mass(mz[i], ppm);
            for (unsigned charge = 1; charge <= max_charge; charge++) {
                float iso = NEUTRON / (float)charge;
                if (std::fabs(delta - iso) <= tol && inten[i] < inten[j]) {
                    if (peaks[i].charge && *peaks[i].charge != charge) continue;
                    peaks[j].intensity += peaks[i].intensity;
                    peaks[j].charge = (uint8_t)charge;
                    peaks[i].charge = (uint8_t)charge;
                    peaks[i].envelope = j;
                }
            }
            j = j == 0 ? 0 : j - 1;
            if (j == 0) break;
        }
    }
    return peaks;
}

void path_compression(std::vector<Deisotoped>& peaks) {  // spectrum.rs:230-239
    for (size_t idx = 0; idx < peaks.size(); idx++) {
        if (peaks[idx].envelope) {
            size_t parent = *peaks[idx].envelope;
            if (peaks[parent].envelope) peaks[idx].envelope = peaks[parent].envelope;
            peaks[idx].intensity = 0.0f;
        }
    }
}

namespace {
struct Peak {  // spectrum.rs:5-25
    float intensity, mass;
};
bool peak_less(const Peak& a, const Peak& b) {
    int c = total_cmp(a.intensity, b.intensity);
    if (c != 0) return c < 0;
    return total_cmp(a.mass, b.mass) < 0;
}
}  // namespace

ProcessedSpectrum SpectrumProcessor::process(const RawSpectrum& spectrum) const {  // spectrum.rs:338-412
    std::vector<Peak> peaks;
    if (spectrum.ms_level == 2) {  // process_ms2, spectrum.rs:279-336
        assert(spectrum.centroid && "profile data");
        uint8_t charge = 3;
        if (!spectrum.precursors.empty() && spectrum.precursors[0].charge)
            charge = *spectrum.precursors[0].charge;
        if (deisotope) {
            std::vector<Deisotoped> d = sage_oracle::deisotope(spectrum.mz.data(), spectrum.intensity.data(),
                                                               spectrum.mz.size(), charge, 10.0f,
                                                               min_deisotope_mz);
            std::stable_sort(d.begin(), d.end(), [](const Deisotoped& a, const Deisotoped& b) {
                int c = total_cmp(b.intensity, a.intensity);
                if (c != 0) return c < 0;
                return total_cmp(a.mz, b.mz) < 0;
            });
            for (const auto& pk : d) {
                if (pk.envelope) continue;
                if (peaks.size() >= take_top_n) break;
                float mass = (pk.mz - PROTON) * (float)(pk.charge ? *pk.charge : 1);
                peaks.push_back({pk.intensity, mass});
            }
        } else {
            for (size_t i = 0; i < spectrum.mz.size(); i++)
                peaks.push_back({spectrum.intensity[i], (spectrum.mz[i] - PROTON) * 1.0f});
            bounded_min_heapify(peaks.data(), peaks.size(), take_top_n, peak_less);
            if (peaks.size() > take_top_n) peaks.resize(take_top_n);
        }
    } else {
        for (size_t i = 0; i < spectrum.mz.size(); i++)
            peaks.push_back({spectrum.intensity[i], (spectrum.mz[i] - PROTON) * 1.0f});
    }
    std::stable_sort(peaks.begin(), peaks.end(),
                     [](const Peak& a, const Peak& b) { return total_cmp(a.mass, b.mass) < 0; });
    ProcessedSpectrum out;
    out.level = spectrum.ms_level;
    out.id = spectrum.id;
    out.file_id = spectrum.file_id;
    out.scan_start_time = spectrum.scan_start_time;
    out.ion_injection_time = spectrum.ion_injection_time;
    out.precursors = spectrum.precursors;
    float tic = 0.0f;
    for (const auto& p : peaks) {
        out.masses.push_back(p.mass);
        out.intensities.push_back(p.intensity);
        tic += p.intensity;
    }
    out.total_ion_current = tic;
    return out;
}

// ---------------------------------------------------------------------------
// scoring.rs
// ---------------------------------------------------------------------------
// f64::ln of the reference (scoring.rs:176, 185, 522).  Rust lowers it to the platform libm's `log`, so what "the reference's
// value" is depends on the libm: glibc < 2.28 rounds it correctly, glibc >= 2.28 (this image: 2.35) is within 0.52 ulp and
// rounds ~99.99 % of this path's arguments correctly.  Two modes:
//   0  the platform libm (std::log) — the reference's arithmetic on THIS platform (default; bench.py's cpu_baseline times it);
//   1  correctly rounded, obtained independently of the product's crlog.h: libquadmath's 113-bit logq rounded to double.
// The product computes the correctly rounded value (sage_amd/csrc/crlog.h); the GPU parity tests hold it to mode 1 bit for bit
// and to mode 0 within 1 ulp with >= 99.9 % equal.
static int g_log_mode = 0;
void set_log_mode(int mode) { g_log_mode = mode; }
int get_log_mode() { return g_log_mode; }
double ln_correctly_rounded(double x) { return (double)logq((__float128)x); }
double ln(double x) { return g_log_mode ? ln_correctly_rounded(x) : std::log(x); }

double lnfact(uint16_t n) {  // scoring.rs:170-177
    if (n == 0) return 1.0;
    double x = (double)n;
    return x * ln(x) - x + 0.5 * ln(x) + 0.5 * ln(M_PI * 2.0 * x);
}

double score_type_score(ScoreType t, uint16_t matched_b, uint16_t matched_y, float summed_b,
                        float summed_y) {  // scoring.rs:179-201
    double score;
    if (t == ScoreType::SageHyperScore) {
        double i = (double)(summed_b + 1.0f) * (double)(summed_y + 1.0f);
        score = ln(i) + lnfact(matched_b) + lnfact(matched_y);
    } else {
        float summed_intensity = summed_b + summed_y;
        score = (double)log1pf(summed_intensity) + lnfact(matched_b) + lnfact(matched_y);
    }
    return std::isfinite(score) ? score : 255.0;
}

uint8_t max_fragment_charge(std::optional<uint8_t> mfc, uint8_t precursor_charge) {  // scoring.rs:239-247
    uint8_t inner = mfc ? (uint8_t)(*mfc + 1) : precursor_charge;
    return std::max<uint8_t>(std::min<uint8_t>(precursor_charge, inner), 2);
}

void Run::matched(size_t index) {  // scoring.rs:780-792
    if (last == index) return;
    if (start + length == index) {
        length += 1;
        longest = std::max(longest, length);
    } else {
        start = index;
        length = 1;
        longest = std::max(longest, length);
    }
    last = index;
}

bool prescore_less(const PreScore& a, const PreScore& b) {  // derived Ord, scoring.rs:43-49
    if (a.matched != b.matched) return a.matched < b.matched;
    if (a.peptide != b.peptide) return a.peptide < b.peptide;
    if (a.precursor_charge != b.precursor_charge) return a.precursor_charge < b.precursor_charge;
    return a.isotope_error < b.isotope_error;
}

void InitialHits::add_assign(InitialHits&& rhs) {  // scoring.rs:60-67
    matched_peaks += rhs.matched_peaks;
    scored_candidates += rhs.scored_candidates;
    preliminary.insert(preliminary.end(), rhs.preliminary.begin(), rhs.preliminary.end());
}

void Scorer::trim_hits(InitialHits& hits) const {  // scoring.rs:322-329
    size_t len = hits.preliminary.size();
    size_t lo = std::min(report_psms * 2, len), hi = len;
    size_t k = std::min(std::max<size_t>(50, lo), hi);  // 50.clamp(lo, hi)
    bounded_min_heapify(hits.preliminary.data(), len, k, prescore_less);
    hits.preliminary.resize(k);
}

InitialHits Scorer::matched_peaks_with_isotope(const ProcessedSpectrum& query, float precursor_mass,
                                               uint8_t precursor_charge, Tolerance precursor_tol,
                                               int8_t isotope_error) const {  // scoring.rs:335-382
    IndexedQuery candidates =
        db->query(precursor_mass - (float)isotope_error * NEUTRON, precursor_tol, fragment_tol);
    if (wc) {
        wc->queries++;
        wc->algorithmic_bytes += 2 * ceil_log2(db->peptides.size() + 1) * 4;
    }
    uint8_t mfc = sage_oracle::max_fragment_charge(max_fragment_charge, precursor_charge);
    size_t potential = candidates.pre_idx_hi - candidates.pre_idx_lo + 1;
    InitialHits hits;
    hits.preliminary.assign(potential, PreScore{});

    for (float peak_mass : query.masses) {
        for (unsigned charge = 1; charge < mfc; charge++) {
            float mass = peak_mass * (float)charge;
            page_search_impl(
                candidates, mass,
                [&](const Theoretical& frag) {
                    size_t idx = (size_t)frag.peptide_index - candidates.pre_idx_lo;
                    PreScore& sc = hits.preliminary[idx];
                    if (sc.matched == 0) {
                        hits.scored_candidates += 1;
                        sc.precursor_charge = precursor_charge;
                        sc.peptide = frag.peptide_index;
                        sc.isotope_error = isotope_error;
                    }
                    sc.matched += 1;
                    hits.matched_peaks += 1;
                },
                wc);
        }
    }
    if (hits.matched_peaks == 0) return hits;
    trim_hits(hits);
    return hits;
}

InitialHits Scorer::matched_peaks(const ProcessedSpectrum& query, float precursor_mass,
                                  uint8_t precursor_charge, Tolerance precursor_tol) const {  // :384-416
    if (min_isotope_err != max_isotope_err) {
        InitialHits hits;
        for (int iso = min_isotope_err; iso <= max_isotope_err; iso++) {
            hits.add_assign(matched_peaks_with_isotope(query, precursor_mass, precursor_charge,
                                                       precursor_tol, (int8_t)iso));
        }
        trim_hits(hits);
        return hits;
    }
    return matched_peaks_with_isotope(query, precursor_mass, precursor_charge, precursor_tol, 0);
}

InitialHits Scorer::initial_hits(const ProcessedSpectrum& query, const Precursor& precursor) const {  // :418-462
    float mz = precursor.mz - PROTON;
    InitialHits hits;
    if (wide_window) {
        for (unsigned z = min_precursor_charge; z <= max_precursor_charge; z++) {
            float precursor_mass = mz * (float)z;
            Tolerance tol = precursor.isolation_window.value_or(Tolerance::Da(-2.4f, 2.4f)).scaled((float)z);
            hits.add_assign(matched_peaks(query, precursor_mass, (uint8_t)z, tol));
        }
    } else if (precursor.charge && !override_precursor_charge) {
        uint8_t charge = *precursor.charge;
        float precursor_mass = mz * (float)charge;
        hits = matched_peaks(query, precursor_mass, charge, precursor_tol);
    } else {
        for (unsigned z = min_precursor_charge; z <= max_precursor_charge; z++) {
            float precursor_mass = mz * (float)z;
            hits.add_assign(matched_peaks(query, precursor_mass, (uint8_t)z, precursor_tol));
        }
    }
    trim_hits(hits);
    return hits;
}

std::pair<Score, std::optional<Fragments>> Scorer::score_candidate(const ProcessedSpectrum& query,
                                                                   const PreScore& pre) const {  // :675-767
    Score score;
    score.peptide = pre.peptide;
    score.precursor_charge = pre.precursor_charge;
    score.isotope_error = pre.isotope_error;
    const Peptide& peptide = db->peptides[score.peptide];
    uint8_t mfc = sage_oracle::max_fragment_charge(max_fragment_charge, score.precursor_charge);
    if (wc) {
        wc->rescored++;
        wc->rescored_residues += peptide.sequence.size();
        wc->algorithmic_bytes += 4 + 5 * peptide.sequence.size();
    }
    Run b_run, y_run;
    Fragments details;
    static thread_local std::vector<float> ions;  // (the reference's IonSeries is an iterator: no allocation per candidate)
    for (Kind kind : db->ion_kinds) {
        ion_series_into(peptide, kind, ions);
        for (size_t idx = 0; idx < ions.size(); idx++) {
            for (unsigned charge = 1; charge < mfc; charge++) {
                float mz = ions[idx] / (float)charge;
                long peak_idx = select_most_intense_peak(query.masses.data(), query.intensities.data(),
                                                         query.masses.size(), mz, fragment_tol, std::nullopt);
                if (peak_idx < 0) continue;
                float peak_mass = query.masses[peak_idx];
                float peak_intensity = query.intensities[peak_idx];
                score.ppm_difference += peak_intensity * std::fabs(mz - peak_mass) * 2E6f / (mz + peak_mass);
                float exp_mz = peak_mass + PROTON;
                float calc_mz = mz + PROTON;
                if (is_nterm_kind(kind)) {
                    score.matched_b += 1;
                    score.summed_b += peak_intensity;
                    b_run.matched(idx);
                } else {
                    score.matched_y += 1;
                    score.summed_y += peak_intensity;
                    y_run.matched(idx);
                }
                if (annotate_matches) {
                    int32_t ord = is_nterm_kind(kind)
                                      ? (int32_t)idx + 1
                                      : (int32_t)(peptide.sequence.empty() ? 0 : peptide.sequence.size() - 1) -
                                            (int32_t)idx;
                    details.kinds.push_back(kind);
                    details.charges.push_back((int32_t)charge);
                    details.mz_experimental.push_back(exp_mz);
                    details.mz_calculated.push_back(calc_mz);
                    details.fragment_ordinals.push_back(ord);
                    details.intensities.push_back(peak_intensity);
                }
            }
        }
    }
    score.hyperscore = score_type_score(score_type, score.matched_b, score.matched_y, score.summed_b, score.summed_y);
    score.longest_b = b_run.longest;
    score.longest_y = y_run.longest;
    score.ppm_difference /= score.summed_b + score.summed_y;
    if (annotate_matches) return {score, std::move(details)};
    return {score, std::nullopt};
}

void Scorer::build_features(const ProcessedSpectrum& query, const Precursor& precursor,
                            const InitialHits& hits, size_t report, std::vector<Feature>& features) const {  // :478-595
    std::vector<std::pair<Score, std::optional<Fragments>>> sv;
    for (const auto& pre : hits.preliminary) {
        if (pre.peptide == 0xFFFFFFFFu) continue;
        auto s = score_candidate(query, pre);
        if ((unsigned)(s.first.matched_b + s.first.matched_y) >= min_matched_peaks) sv.push_back(std::move(s));
    }
    std::stable_sort(sv.begin(), sv.end(), [](const auto& a, const auto& b) {
        return total_cmp(b.first.hyperscore, a.first.hyperscore) < 0;
    });
    double lambda = (double)hits.matched_peaks / (double)hits.scored_candidates;
    float mz = precursor.mz - PROTON;

    for (size_t idx = 0; idx < std::min(report, sv.size()); idx++) {
        const Score& score = sv[idx].first;
        const Peptide& peptide = db->peptides[score.peptide];
        float precursor_mass = mz * (float)score.precursor_charge;
        double next = idx + 1 < sv.size() ? sv[idx + 1].first.hyperscore : 0.0;
        double best = sv[0].first.hyperscore;
        uint16_t k = (uint16_t)(score.matched_b + score.matched_y);
        double log10_poisson = ((double)k * ln(lambda) - lambda - lnfact(k)) / M_LN10;
        float isotope_error = (float)score.isotope_error * NEUTRON;
        float delta_mass = (precursor_mass - peptide.monoisotopic - isotope_error) * 2E6f /
                           (precursor_mass - isotope_error + peptide.monoisotopic);
        Feature f;
        f.peptide_idx = score.peptide;
        f.file_id = query.file_id;
        f.rank = (uint32_t)idx + 1;
        f.label = peptide.label();
        f.expmass = precursor_mass;
        f.calcmass = peptide.monoisotopic;
        f.charge = score.precursor_charge;
        f.rt = query.scan_start_time;
        f.ims = query.precursors.front().inverse_ion_mobility.value_or(0.0f);
        f.delta_mass = delta_mass;
        f.isotope_error = isotope_error;
        f.average_ppm = score.ppm_difference;
        f.hyperscore = score.hyperscore;
        f.delta_next = score.hyperscore - next;
        f.delta_best = best - score.hyperscore;
        f.matched_peaks = k;
        f.matched_intensity_pct = 100.0f * (score.summed_b + score.summed_y) / query.total_ion_current;
        f.poisson = std::isfinite(log10_poisson) ? log10_poisson : -std::numeric_limits<double>::infinity();
        f.longest_b = (uint32_t)score.longest_b;
        f.longest_y = (uint32_t)score.longest_y;
        f.longest_y_pct = (float)score.longest_y / (float)peptide.sequence.size();
        f.peptide_len = peptide.sequence.size();
        f.scored_candidates = (uint32_t)hits.scored_candidates;
        f.missed_cleavages = peptide.missed_cleavages;
        f.ms2_intensity = score.summed_b + score.summed_y;
        f.fragments = std::move(sv[idx].second);
        if (wc) {
            wc->reported++;
            wc->algorithmic_bytes += 64;
        }
        features.push_back(std::move(f));
    }
}

std::vector<Feature> Scorer::score_standard(const ProcessedSpectrum& query) const {  // :465-474
    assert(!query.precursors.empty() && "missing MS1 precursor");
    const Precursor& precursor = query.precursors.front();
    InitialHits hits = initial_hits(query, precursor);
    std::vector<Feature> features;
    build_features(query, precursor, hits, report_psms, features);
    return features;
}

void Scorer::remove_matched_peaks(ProcessedSpectrum& query, const Feature& psm) const {  // :598-644
    const Peptide& peptide = db->peptides[psm.peptide_idx];
    uint8_t mfc = sage_oracle::max_fragment_charge(max_fragment_charge, psm.charge);
    std::vector<std::pair<float, float>> to_remove;
    for (Kind kind : db->ion_kinds) {
        for (float frag : ion_series(peptide, kind)) {
            for (unsigned charge = 1; charge < mfc; charge++) {
                long peak_idx = select_most_intense_peak(query.masses.data(), query.intensities.data(),
                                                         query.masses.size(), frag / (float)charge,
                                                         fragment_tol, std::nullopt);
                if (peak_idx >= 0) to_remove.emplace_back(query.masses[peak_idx], query.intensities[peak_idx]);
            }
        }
    }
    std::vector<float> masses, intensities;
    for (size_t idx = 0; idx < query.masses.size(); idx++) {
        bool found = false;
        for (const auto& pk : to_remove)
            if (pk.first == query.masses[idx] && pk.second == query.intensities[idx]) { found = true; break; }
        if (!found) {
            masses.push_back(query.masses[idx]);
            intensities.push_back(query.intensities[idx]);
        }
    }
    query.masses = std::move(masses);
    query.intensities = std::move(intensities);
    float tic = 0.0f;
    for (float x : query.intensities) tic += x;
    query.total_ion_current = tic;
}

std::vector<Feature> Scorer::score_chimera_fast(const ProcessedSpectrum& query_in) const {  // :648-672
    assert(!query_in.precursors.empty() && "missing MS1 precursor");
    const Precursor precursor = query_in.precursors.front();
    ProcessedSpectrum query = query_in;
    InitialHits hits = initial_hits(query, precursor);
    std::vector<Feature> candidates;
    size_t prev = 0;
    while (candidates.size() < report_psms) {
        build_features(query, precursor, hits, 1, candidates);
        if (candidates.size() > prev) {
            remove_matched_peaks(query, candidates[prev]);
            candidates[prev].rank = (uint32_t)prev + 1;
            prev = candidates.size();
        } else {
            break;
        }
    }
    return candidates;
}

std::vector<Feature> Scorer::score(const ProcessedSpectrum& query) const {  // :300-309
    assert(query.level == 2 && "internal bug, trying to score a non-MS2 scan!");
    if (wc) {
        wc->peaks += query.masses.size();
        wc->algorithmic_bytes += 8 * query.masses.size();
    }
    return chimera ? score_chimera_fast(query) : score_standard(query);
}

void Scorer::quick_score(const ProcessedSpectrum& query, bool prefilter_low_memory,
                         std::vector<uint8_t>& keep) const {  // scoring.rs:255-298
    assert(query.level == 2);
    const Precursor& precursor = query.precursors.front();
    InitialHits hits = initial_hits(query, precursor);
    if (prefilter_low_memory) {
        std::vector<Score> sv;
        for (const auto& pre : hits.preliminary) {
            if (pre.peptide == 0xFFFFFFFFu) continue;
            Score s = score_candidate(query, pre).first;
            if ((unsigned)(s.matched_b + s.matched_y) < min_matched_peaks) continue;
            sv.push_back(s);
        }
        size_t k = std::min(report_psms, sv.size());
        // heap.rs uses `<`/`>` => the *derived* PartialOrd of Score (field order, peptide first),
        // not the hyperscore Ord (scoring.rs:17 vs :34-40)
        auto less = [](const Score& a, const Score& b) {
            if (a.peptide != b.peptide) return a.peptide < b.peptide;
            if (a.matched_b != b.matched_b) return a.matched_b < b.matched_b;
            if (a.matched_y != b.matched_y) return a.matched_y < b.matched_y;
            if (a.summed_b != b.summed_b) return a.summed_b < b.summed_b;
            if (a.summed_y != b.summed_y) return a.summed_y < b.summed_y;
            if (a.longest_b != b.longest_b) return a.longest_b < b.longest_b;
            if (a.longest_y != b.longest_y) return a.longest_y < b.longest_y;
            if (a.hyperscore != b.hyperscore) return a.hyperscore < b.hyperscore;
            if (a.ppm_difference != b.ppm_difference) return a.ppm_difference < b.ppm_difference;
            if (a.precursor_charge != b.precursor_charge) return a.precursor_charge < b.precursor_charge;
            return a.isotope_error < b.isotope_error;
        };
        bounded_min_heapify(sv.data(), sv.size(), k, less);
        for (size_t i = 0; i < k; i++) keep[sv[i].peptide] = 1;
    } else {
        for (const auto& pre : hits.preliminary)
            if (pre.peptide != 0xFFFFFFFFu) keep[pre.peptide] = 1;
    }
}

std::vector<Score> Scorer::brute_force_scores(const ProcessedSpectrum& query, float precursor_mass,
                                              uint8_t charge, Tolerance precursor_tol,
                                              int8_t isotope_error) const {
    // Not in the reference.  Every peptide with mono in bounds(precursor_mass - iso*NEUTRON) is
    // scored directly; used to cross-check index + k-select where k-select does not bind.
    auto b = precursor_tol.bounds(precursor_mass - (float)isotope_error * NEUTRON);
    std::vector<Score> out;
    for (size_t i = 0; i < db->peptides.size(); i++) {
        float m = db->peptides[i].monoisotopic;
        if (m >= b.first && m <= b.second) {
            PreScore pre;
            pre.peptide = (uint32_t)i;
            pre.precursor_charge = charge;
            pre.isotope_error = isotope_error;
            out.push_back(score_candidate(query, pre).first);
        }
    }
    return out;
}

}  // namespace sage_oracle
