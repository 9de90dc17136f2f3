// ORACLE — TEST INFRASTRUCTURE ONLY.
// Pins the restatement against the reference's own unit-test vectors (each CHECK cites the
// reference test it restates).  Run by tests/test_oracle_golden.py; exit code = #failures.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "sage_oracle.hpp"
#include "../sage_amd/csrc/detmath.h"  // the PRODUCT's elementary functions: checked here, never used by the checker itself
#include "detmath_oracle.h"            // the checker's own copy
#include <cstring>

using namespace sage_oracle;

static int failures = 0, checks = 0;
#define CHECK(cond)                                                             \
    do {                                                                        \
        checks++;                                                               \
        if (!(cond)) {                                                          \
            failures++;                                                         \
            std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);         \
        }                                                                       \
    } while (0)

static Peptide peptide_of(const std::string& s) {
    Digest d;
    d.sequence = s;
    Peptide p;
    bool ok = Peptide::from_digest(d, p);
    CHECK(ok);
    return p;
}

static std::vector<std::string> seqs(const std::vector<Digest>& v) {
    std::vector<std::string> out;
    for (auto& d : v) out.push_back(d.sequence);
    return out;
}

static EnzymeParameters enz(size_t min_len, size_t max_len, uint8_t mc, const char* cleave, const char* skip,
                            bool cterm, bool semi = false) {
    EnzymeParameters e;
    e.min_len = min_len;
    e.max_len = max_len;
    e.missed_cleavages = mc;
    e.enzyme = Enzyme::make(cleave, skip, cterm, semi);
    return e;
}

static void test_mass() {  // mass.rs:143-157
    auto b = Tolerance::Ppm(-10.0f, 20.0f).bounds(1000.0f);
    CHECK(b.first == 999.99f && b.second == 1000.02f);
    b = Tolerance::Ppm(-10.0f, 10.0f).bounds(487.0f);
    CHECK(b.first == 486.99513f && b.second == 487.00487f);
    b = Tolerance::Ppm(-50.0f, 50.0f).bounds(1000.0f);
    CHECK(b.first == 999.95f && b.second == 1000.05f);
    for (const char* c = "ACDEFGHIKLMNPQRSTVWYUO"; *c; c++) CHECK(monoisotopic((uint8_t)*c) > 0.0f);  // mass.rs:136-141
}

static void test_binary_search() {  // database.rs:569-593
    double data[] = {1.0, 1.5, 2.0, 2.5, 3.0, 3.5, 4.0};
    auto key = [](const double& a, const double& b) { return total_cmp(a, b); };
    auto r = binary_search_slice(data, 7, key, 1.75, 3.5);
    CHECK(r.first == 1 && r.second == 6);
    r = binary_search_slice(data, 7, key, 0.0, 5.0);
    CHECK(r.first == 0 && r.second == 7);
    double run[] = {1.0, 1.5, 1.5, 1.5, 1.5, 2.0, 2.5, 3.0, 3.0, 3.5, 4.0};
    r = binary_search_slice(run, 11, key, 1.5, 3.25);
    CHECK(run[r.first] <= 1.5 && run[r.second] > 3.25);
    CHECK(r.first == 0 && r.second == 9);
}

static void test_scoring_units() {  // scoring.rs:799-830
    Run run;
    run.matched(1); run.matched(2); run.matched(3); run.matched(3); run.matched(3);
    CHECK(run.length == 3 && run.longest == 3);
    run.matched(5); run.matched(5);
    CHECK(run.length == 1 && run.longest == 3);
    run.matched(6);
    CHECK(run.length == 2);
    CHECK(max_fragment_charge(std::nullopt, 1) == 2);
    CHECK(max_fragment_charge(std::nullopt, 2) == 2);
    CHECK(max_fragment_charge(std::nullopt, 3) == 3);
    CHECK(max_fragment_charge(std::nullopt, 4) == 4);
    CHECK(max_fragment_charge(uint8_t(1), 2) == 2);
    CHECK(max_fragment_charge(uint8_t(1), 3) == 2);
    CHECK(max_fragment_charge(uint8_t(2), 4) == 3);
    CHECK(max_fragment_charge(uint8_t(4), 1) == 2);
    CHECK(lnfact(0) == 1.0);  // the reference's quirk, scoring.rs:171-172
}

static bool within(const std::vector<float>& obs, const std::vector<float>& exp, float charge, float add = 0.f) {
    if (obs.size() != exp.size()) return false;
    for (size_t i = 0; i < obs.size(); i++) {
        float mz = (obs[i] + charge * PROTON) / charge;
        if (!(std::fabs(exp[i] + add - mz) < 0.005f)) return false;
    }
    return true;
}

static void test_ion_series() {  // ion_series.rs:129-328
    Peptide p = peptide_of("PEPTIDE");
    CHECK(within(ion_series(p, Kind::A), {70.065f, 199.108f, 296.160f, 397.208f, 510.292f, 625.32f}, 1));
    CHECK(within(ion_series(p, Kind::B), {98.0600f, 227.1026f, 324.155f, 425.2030f, 538.287f, 653.314f}, 1));
    CHECK(within(ion_series(p, Kind::C), {115.086f, 244.129f, 341.182f, 442.229f, 555.314f, 670.341f}, 1));
    CHECK(within(ion_series(p, Kind::X), {729.294f, 600.251f, 503.198f, 402.151f, 289.066f, 174.039f}, 1));
    CHECK(within(ion_series(p, Kind::Y), {703.314f, 574.2719f, 477.219f, 376.171f, 263.0874f, 148.0604f}, 1));
    CHECK(within(ion_series(p, Kind::Z), {686.288f, 557.245f, 460.193f, 359.145f, 246.061f, 131.034f}, 1));
    CHECK(within(ion_series(p, Kind::Y), {352.16087f, 287.6396f, 239.11319f, 188.58935f, 132.04732f, 74.53385f}, 2));
    Peptide q = peptide_of("EDITPEP");
    CHECK(within(ion_series(q, Kind::Y), {336.16596f, 278.6525f, 222.11046f, 171.58662f, 123.060237f, 58.53894f}, 2));
    // nterm_mod / cterm_mod / internal_mod
    ModSpec pn{ModSpec::PeptideN, -1}, pc{ModSpec::PeptideC, -1}, ri{ModSpec::Residue, 'I'};
    std::vector<float> eb = {98.06004f, 227.10263f, 324.1554f, 425.20306f, 538.2872f, 653.3141f};
    std::vector<float> ey = {703.31447f, 574.27188f, 477.21912f, 376.17144f, 263.08737f, 148.06043f};
    Peptide n = p.apply({}, {{pn, 229.01f}}, 1)[0];
    CHECK(within(ion_series(n, Kind::B), eb, 1, 229.01f));
    CHECK(within(ion_series(n, Kind::Y), ey, 1));
    Peptide c = p.apply({}, {{pc, 229.01f}}, 1)[0];
    CHECK(std::fabs(c.monoisotopic - 1028.37f) < 0.001f);
    CHECK(within(ion_series(c, Kind::B), eb, 1));
    CHECK(within(ion_series(c, Kind::Y), ey, 1, 229.01f));
    Peptide im = p.apply({}, {{ri, 29.0f}}, 1)[0];
    std::vector<float> ebi = eb, eyi = ey;
    ebi[4] += 29.0f; ebi[5] += 29.0f;
    for (int i = 0; i < 4; i++) eyi[i] += 29.0f;
    CHECK(within(ion_series(im, Kind::B), ebi, 1));
    CHECK(within(ion_series(im, Kind::Y), eyi, 1));
}

static void test_heap() {  // heap.rs:62-100
    std::mt19937 rng(7);
    auto less = [](int a, int b) { return a < b; };
    for (int trial = 0; trial < 300; trial++) {
        size_t n = rng() % 200;
        size_t k = rng() % 80;
        std::vector<int> data(n);
        for (auto& x : data) x = (int)(rng() % 50) - 25;
        if (trial == 0) { data.resize(500); for (int i = 0; i < 500; i++) data[i] = i; k = 50; n = 500; }
        if (trial == 1) { data.resize(500); for (int i = 0; i < 500; i++) data[i] = 499 - i; k = 50; n = 500; }
        k = std::min(k, data.size());
        std::vector<int> sorted = data;
        std::stable_sort(sorted.begin(), sorted.end(), [](int a, int b) { return a > b; });
        bounded_min_heapify(data.data(), data.size(), k, less);
        bool heap_ok = true;
        for (size_t i = 1; i < k; i++) if (data[(i - 1) / 2] > data[i]) heap_ok = false;
        CHECK(heap_ok || k == data.size());
        std::vector<int> top(data.begin(), data.begin() + k);
        std::stable_sort(top.begin(), top.end(), [](int a, int b) { return a > b; });
        CHECK(top == std::vector<int>(sorted.begin(), sorted.begin() + k));
    }
}

static void test_spectrum() {  // spectrum.rs:419-605
    float mz[] = {800.9f, 800.9f + NEUTRON * 1.0f, 800.9f + NEUTRON * 2.0f, 803.4080f, 804.4108f,
                  805.4106f, 806.4116f, 810.0f, 812.0f, 812.0f + NEUTRON / 2.0f};
    float in[] = {2.f, 1.5f, 1.f, 4.f, 3.f, 2.f, 1.f, 1.f, 9.0f, 4.5f};
    auto pk = deisotope(mz, in, 10, 2, 5.0f, 800.91f);
    struct E { float inten; int charge; int env; };
    E exp1[] = {{2.0f, -1, -1}, {2.5f, 1, -1}, {1.0f, 1, 1}, {10.0f, 1, -1}, {6.0f, 1, 3},
                {3.0f, 1, 4},   {1.0f, 1, 5},  {1.0f, -1, -1}, {13.5f, 2, -1}, {4.5f, 2, 8}};
    for (int i = 0; i < 10; i++) {
        CHECK(pk[i].mz == mz[i]);
        CHECK(pk[i].intensity == exp1[i].inten);
        CHECK((pk[i].charge ? (int)*pk[i].charge : -1) == exp1[i].charge);
        CHECK((pk[i].envelope ? (int)*pk[i].envelope : -1) == exp1[i].env);
    }
    path_compression(pk);
    E exp2[] = {{2.0f, -1, -1}, {2.5f, 1, -1}, {0.0f, 1, 1}, {10.0f, 1, -1}, {0.0f, 1, 3},
                {0.0f, 1, 3},   {0.0f, 1, 3},  {1.0f, -1, -1}, {13.5f, 2, -1}, {0.0f, 2, 8}};
    for (int i = 0; i < 10; i++) {
        CHECK(pk[i].intensity == exp2[i].inten);
        CHECK((pk[i].envelope ? (int)*pk[i].envelope : -1) == exp2[i].env);
    }
    {
        float m[] = {99.0f, 100.0f, 100.01f, 100.02f, 101.0f};
        float it[] = {10.0f, 20.0f, 50.0f, 30.0f, 100.0f};
        CHECK(select_most_intense_peak(m, it, 5, 100.01f, Tolerance::Da(-0.02f, 0.02f), std::nullopt) == 2);
    }
    {
        float label = 126.127726f;
        float m[] = {label - PROTON - 0.01f, label - PROTON, label - PROTON + 0.01f};
        float it[] = {10.0f, 100.0f, 50.0f};
        CHECK(select_most_intense_peak(m, it, 3, label, Tolerance::Da(-0.005f, 0.005f), -PROTON) == 1);
    }
    {  // process_ms1_without_mobility...
        SpectrumProcessor sp;
        sp.take_top_n = 10; sp.deisotope = false; sp.min_deisotope_mz = 0.0f;
        RawSpectrum raw;
        raw.ms_level = 1; raw.file_id = 7;
        raw.mz = {102.0f, 100.0f, 101.0f};
        raw.intensity = {30.0f, 10.0f, 20.0f};
        auto out = sp.process(raw);
        CHECK(out.file_id == 7);
        CHECK(out.masses == (std::vector<float>{100.0f - PROTON, 101.0f - PROTON, 102.0f - PROTON}));
        CHECK(out.intensities == (std::vector<float>{10.0f, 20.0f, 30.0f}));
        CHECK(out.total_ion_current == 60.0f);
    }
}

static void test_enzyme() {  // enzyme.rs:401-700
    const std::string s = "MADEEKLPPGWEKRMSRSSGRVYYFNHITNASQWERPSGN";
    {
        auto d = enz(2, 50, 0, "KR", "P", true).digest(s, "");
        CHECK(seqs(d) == (std::vector<std::string>{"MADEEK", "LPPGWEK", "MSR", "SSGR", "VYYFNHITNASQWERPSGN"}));
        CHECK(d[0].position == Position::Nterm && d[1].position == Position::Internal &&
              d[4].position == Position::Cterm);
    }
    CHECK(seqs(enz(0, 50, 1, "KR", "P", true).digest(s, "")) ==
          (std::vector<std::string>{"MADEEK", "LPPGWEK", "R", "MSR", "SSGR", "VYYFNHITNASQWERPSGN",
                                    "MADEEKLPPGWEK", "LPPGWEKR", "RMSR", "MSRSSGR", "SSGRVYYFNHITNASQWERPSGN"}));
    CHECK(seqs(enz(0, 50, 2, "KR", "P", true).digest(s, "")) ==
          (std::vector<std::string>{"MADEEK", "LPPGWEK", "R", "MSR", "SSGR", "VYYFNHITNASQWERPSGN",
                                    "MADEEKLPPGWEK", "LPPGWEKR", "RMSR", "MSRSSGR", "SSGRVYYFNHITNASQWERPSGN",
                                    "MADEEKLPPGWEKR", "LPPGWEKRMSR", "RMSRSSGR", "MSRSSGRVYYFNHITNASQWERPSGN"}));
    CHECK(seqs(enz(2, 50, 0, "KR", "", true).digest(s, "")) ==
          (std::vector<std::string>{"MADEEK", "LPPGWEK", "MSR", "SSGR", "VYYFNHITNASQWER", "PSGN"}));
    const std::string w = s + "W";
    CHECK(seqs(enz(1, 50, 0, "D", "", false).digest(w, "")) ==
          (std::vector<std::string>{"MA", "DEEKLPPGWEKRMSRSSGRVYYFNHITNASQWERPSGNW"}));
    CHECK(seqs(enz(1, 50, 0, "FYWL", "", true).digest(w, "")) ==
          (std::vector<std::string>{"MADEEKL", "PPGW", "EKRMSRSSGRVY", "Y", "F", "NHITNASQW", "ERPSGNW"}));
    {
        std::vector<std::string> exp;
        for (size_t i = 0; i + 5 <= w.size(); i++) exp.push_back(w.substr(i, 5));
        EnzymeParameters e; e.min_len = 5; e.max_len = 5; e.missed_cleavages = 0;  // enzyme: None
        CHECK(seqs(e.digest(w, "")) == exp);
    }
    {
        std::vector<std::string> exp;
        for (size_t win = 5; win <= 7; win++)
            for (size_t i = 0; i + win <= w.size(); i++) exp.push_back(w.substr(i, win));
        CHECK(seqs(enz(5, 7, 0, "", "", true).digest(w, "")) == exp);
    }
    CHECK(seqs(enz(0, (size_t)-1, 0, "$", "", true).digest(w, "")) == (std::vector<std::string>{w}));
}

static std::vector<std::string> var_mod_sequence(const Peptide& p, const std::vector<std::pair<ModSpec, float>>& mods,
                                                 size_t combo) {
    std::vector<std::string> out;
    for (auto& q : p.apply(mods, {}, combo)) out.push_back(q.to_string());
    return out;
}

static void test_peptide() {  // peptide.rs:429-720
    ModSpec M{ModSpec::Residue, 'M'}, C{ModSpec::Residue, 'C'}, S{ModSpec::Residue, 'S'};
    ModSpec PN{ModSpec::PeptideN, -1}, PC{ModSpec::PeptideC, -1}, QN{ModSpec::ProteinN, -1}, QC{ModSpec::ProteinC, -1};
    {
        auto d = enz(0, 50, 0, "KR", "P", true).digest("MPEPTIDEKMSAGEKEND", "");
        CHECK(d.size() == 3);
        std::vector<Peptide> peps(3);
        for (int i = 0; i < 3; i++) CHECK(Peptide::from_digest(d[i], peps[i]));
        CHECK(peps[0].to_string() == "MPEPTIDEK" && peps[0].position == Position::Nterm);
        CHECK(peps[1].to_string() == "MSAGEK" && peps[1].position == Position::Internal);
        CHECK(peps[2].to_string() == "END" && peps[2].position == Position::Cterm);
        std::vector<std::pair<ModSpec, float>> mods = {{QN, 42.0f}, {QC, 11.0f}, {PN, 12.0f}, {PC, 19.0f}};
        CHECK(var_mod_sequence(peps[0], mods, 2) ==
              (std::vector<std::string>{"MPEPTIDEK", "[+42]-MPEPTIDEK", "[+12]-MPEPTIDEK", "MPEPTIDEK-[+19]",
                                        "[+42]-MPEPTIDEK-[+19]", "[+12]-MPEPTIDEK-[+19]"}));
        CHECK(var_mod_sequence(peps[1], mods, 2) ==
              (std::vector<std::string>{"MSAGEK", "[+12]-MSAGEK", "MSAGEK-[+19]", "[+12]-MSAGEK-[+19]"}));
        CHECK(var_mod_sequence(peps[2], mods, 2) ==
              (std::vector<std::string>{"END", "END-[+11]", "[+12]-END", "END-[+19]", "[+12]-END-[+11]",
                                        "[+12]-END-[+19]"}));
    }
    Peptide g = peptide_of("GCMGCMG");
    CHECK(var_mod_sequence(g, {{M, 16.0f}, {C, 57.0f}}, 2) ==
          (std::vector<std::string>{"GCMGCMG", "GCM[+16]GCMG", "GCMGCM[+16]G", "GC[+57]MGCMG", "GCMGC[+57]MG",
                                    "GCM[+16]GCM[+16]G", "GC[+57]M[+16]GCMG", "GCM[+16]GC[+57]MG",
                                    "GC[+57]MGCM[+16]G", "GCMGC[+57]M[+16]G", "GC[+57]MGC[+57]MG"}));
    CHECK(var_mod_sequence(peptide_of("AAAAAAAA"), {{M, 16.0f}, {C, 57.0f}}, 2) ==
          (std::vector<std::string>{"AAAAAAAA"}));
    CHECK(var_mod_sequence(g, {{PN, 42.0f}, {M, 16.0f}}, 3) ==
          (std::vector<std::string>{"GCMGCMG", "[+42]-GCMGCMG", "GCM[+16]GCMG", "GCMGCM[+16]G",
                                    "[+42]-GCM[+16]GCMG", "[+42]-GCMGCM[+16]G", "GCM[+16]GCM[+16]G",
                                    "[+42]-GCM[+16]GCM[+16]G"}));
    CHECK(var_mod_sequence(g, {{PC, 42.0f}, {M, 16.0f}}, 3) ==
          (std::vector<std::string>{"GCMGCMG", "GCMGCMG-[+42]", "GCM[+16]GCMG", "GCMGCM[+16]G",
                                    "GCM[+16]GCMG-[+42]", "GCMGCM[+16]G-[+42]", "GCM[+16]GCM[+16]G",
                                    "GCM[+16]GCM[+16]G-[+42]"}));
    CHECK(var_mod_sequence(peptide_of("GGGSGGGS"), {{S, 79.0f}, {S, 541.0f}}, 2) ==
          (std::vector<std::string>{"GGGSGGGS", "GGGS[+79]GGGS", "GGGSGGGS[+79]", "GGGS[+541]GGGS",
                                    "GGGSGGGS[+541]", "GGGS[+79]GGGS[+79]", "GGGS[+79]GGGS[+541]",
                                    "GGGS[+541]GGGS[+79]", "GGGS[+541]GGGS[+541]"}));
    {  // apply_mods: variable then static
        std::vector<std::string> out;
        for (auto& q : peptide_of("AACAACAA").apply({{C, 30.0f}}, {{C, 57.0f}}, 2)) out.push_back(q.to_string());
        CHECK(out == (std::vector<std::string>{"AAC[+57]AAC[+57]AA", "AAC[+30]AAC[+57]AA", "AAC[+57]AAC[+30]AA",
                                                "AAC[+30]AAC[+30]AA"}));
    }
    {  // test_psuedo_forward
        for (auto& d : enz(3, 30, 0, "KR", "P", true).digest("MADEEKLPPGWEKRMSRSSGRVYYFNHITNASQWERPSGN", "")) {
            Peptide fwd, rev;
            CHECK(Peptide::from_digest(d, fwd));
            rev = fwd.reverse();
            CHECK(!fwd.decoy && rev.decoy);
            CHECK(fwd.sequence.size() < 4 || fwd.sequence != rev.sequence);
            CHECK(rev.reverse().to_string() == fwd.to_string());
        }
    }
}

static const char* FASTA_Q99536 =
    ">sp|Q99536|VAT1_HUMAN Synaptic vesicle membrane protein VAT-1 homolog OS=Homo sapiens OX=9606 GN=VAT1 PE=1 SV=2\n"
    "MSDEREVAEAATGEDASSPPPKTEAASDPQHPAASEGAAAAAASPPLLRCLVLTGFGGYD\n"
    "KVKLQSRPAAPPAPGPGQLTLRLRACGLNFADLMARQGLYDRLPPLPVTPGMEGAGVVIA\n"
    "VGEGVSDRKAGDRVMVLNRSGMWQEEVTVPSVQTFLIPEAMTFEEAAALLVNYITAYMVL\n"
    "FDFGNLQPGHSVLVHMAAGGVGMAAVQLCRTVENVTVFGTASASKHEALKENGVTHPIDY\n"
    "HTTDYVDEIKKISPKGVDIVMDPLGGSDTAKGYNLLKPMGKVVTYGMANLLTGPKRNLMA\n"
    "LARTWWNQFSVTALQLLQANRAVCGFHLGYLDGEVELVSGVVARLLALYNQGHIKPHIDS\n"
    "VWPFEKVADAMKQMQEKKNVGKVLLVPGPEKEN\n";

static void test_database() {
    {  // database.rs:595-671 `digestion`
        std::string fasta =
            "\n        >sp|AAAAA\n        MEWKLEQSMREQALLKAQLTQLK\n        >sp|BBBBB\n        RMEWKLEQSMREQALLKAQLTQLK\n        ";
        Fasta f = Fasta::parse(fasta, "rev_", false);
        CHECK(f.targets.size() == 2);
        CHECK(f.targets[0].first == "sp|AAAAA" && f.targets[0].second == "MEWKLEQSMREQALLKAQLTQLK");
        CHECK(f.targets[1].first == "sp|BBBBB" && f.targets[1].second == "RMEWKLEQSMREQALLKAQLTQLK");
        Parameters P;
        P.bucket_size = 128;
        P.enzyme.missed_cleavages = 1;
        P.enzyme.min_len = 6;
        P.enzyme.max_len = 10;
        P.peptide_min_mass = 150.0f;
        P.peptide_max_mass = 5000.0f;
        P.variable_mods = {{ModSpec{ModSpec::ProteinN, -1}, {42.0f}}};
        P.max_variable_mods = 2;
        P.generate_decoys = false;
        auto peps = P.digest(f);
        std::vector<std::string> got;
        for (auto& p : peps) got.push_back(p.to_string());
        CHECK(got == (std::vector<std::string>{"EQALLK", "LEQSMR", "AQLTQLK", "MEWKLEQSMR", "[+42]-MEWKLEQSMR"}));
        for (int i = 0; i < 4 && i < (int)peps.size(); i++) CHECK(peps[i].proteins.size() == 2);
        CHECK(!peps.empty() && peps.back().proteins == std::vector<std::string>{"sp|AAAAA"});
    }
    {  // crates/sage/tests/integration.rs:30-70 `check_all_ions_visited`
        std::mt19937 rng(11);
        for (int trial = 0; trial < 60; trial++) {
            size_t bucket = 1 + rng() % 8192;
            if (trial < 4) bucket = (size_t[]){1, 2, 8192, 64}[trial];
            size_t pow2 = 1;
            while (pow2 < bucket) pow2 <<= 1;
            Parameters P;  // Builder{bucket_size, ..Default}
            P.bucket_size = pow2;
            Fasta f = Fasta::parse(FASTA_Q99536, "rev_", false);
            IndexedDatabase db = P.build(f);
            float target = (float)(rng() % 300000) / 100.0f;
            if (trial % 7 == 0) target = -50.0f;
            Tolerance ftol = Tolerance::Da(-100.0f, 100.0f);
            auto fb = ftol.bounds(target);
            std::vector<size_t> expected(db.peptides.size(), 0), visited(db.peptides.size(), 0);
            bool inv = true;
            for (size_t c = 0; c * db.bucket_size < db.fragments.size(); c++) {
                uint32_t last = 0;
                size_t e = std::min((c + 1) * db.bucket_size, db.fragments.size());
                for (size_t i = c * db.bucket_size; i < e; i++) {
                    const auto& fr = db.fragments[i];
                    if (fr.peptide_index < last) inv = false;
                    if (!(fr.fragment_mz >= db.min_value[c])) inv = false;
                    if (c + 1 < db.min_value.size() && !(fr.fragment_mz <= db.min_value[c + 1])) inv = false;
                    if (fr.fragment_mz >= fb.first && fr.fragment_mz <= fb.second) expected[fr.peptide_index]++;
                    last = fr.peptide_index;
                }
            }
            CHECK(inv);
            auto q = db.query(1000.0f, Tolerance::Da(-5000.0f, 5000.0f), ftol);
            q.page_search(target, [&](const Theoretical& fr) { visited[fr.peptide_index]++; });
            CHECK(expected == visited);
        }
    }
}

// The elementary functions of the rescoring contract (sage_amd/csrc/detmath.h) against the platform libm — what Rust's
// f64::ln_1p / exp / f32::ln_1p call: never more than 1 ulp apart over the ranges the rescoring uses; special values equal.
static long ulps(double a, double b) {
    if (a == b || (a != a && b != b)) return 0;
    int64_t x, y;
    std::memcpy(&x, &a, 8);
    std::memcpy(&y, &b, 8);
    if ((x < 0) != (y < 0)) return 1L << 40;
    return std::labs((long)(x - y));
}
static void test_detmath() {
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> e(-60.0, 12.0), u(-745.0, 709.0);
    long worst_l = 0, worst_e = 0, worst_f = 0;
    for (int i = 0; i < 2000000; i++) {
        double x = std::exp2(e(rng));
        if (i & 1) x = -x;
        if (x <= -1.0) x = -0.999999 * std::generate_canonical<double, 53>(rng);
        worst_l = std::max(worst_l, ulps(sagedet::det_log1p(x), std::log1p(x)));
        const double y = (i % 3 == 0) ? u(rng) : -std::exp2(e(rng));
        worst_e = std::max(worst_e, ulps(sagedet::det_exp(y), std::exp(y)));
        // the checker's copy and the product's header are the same functions, bit for bit
        CHECK(ulps(orcdet::det_log1p(x), sagedet::det_log1p(x)) == 0 && ulps(orcdet::det_exp(y), sagedet::det_exp(y)) == 0);
        const float xf = (float)std::fabs(x);
        const float a = sagedet::det_log1pf(xf), b = std::log1p(xf);
        worst_f = std::max(worst_f, a == b ? 0L : (std::nextafterf(a, b) == b ? 1L : 2L));
    }
    std::printf("detmath vs libm: log1p <= %ld ulp, exp <= %ld ulp, log1pf <= %ld ulp(f32)\n", worst_l, worst_e, worst_f);
    CHECK(worst_l <= 1);
    CHECK(worst_e <= 1);
    CHECK(worst_f <= 1);
    const double inf = std::numeric_limits<double>::infinity();
    CHECK(sagedet::det_log1p(0.0) == 0.0 && sagedet::det_log1p(-1.0) == -inf && sagedet::det_log1p(inf) == inf);
    CHECK(sagedet::det_log1p(-2.0) != sagedet::det_log1p(-2.0));  // NaN
    CHECK(sagedet::det_exp(0.0) == 1.0 && sagedet::det_exp(-inf) == 0.0 && sagedet::det_exp(inf) == inf && sagedet::det_exp(-800.0) == 0.0);
    CHECK(sagedet::det_exp(1.0) == std::nextafter(std::exp(1.0), 3.0) || sagedet::det_exp(1.0) == std::exp(1.0));
    // the blocked order is the sequential order up to DET_BLOCK elements
    std::vector<double> v(sagedet::DET_BLOCK);
    for (auto& x : v) x = std::generate_canonical<double, 53>(rng) * 1e3;
    double seq = 0.0;
    for (double x : v) seq += x;
    CHECK(sagedet::blocked_sum(v.size(), [&](uint64_t i) { return v[i]; }) == seq);
}

int main() {
    test_detmath();
    test_mass();
    test_binary_search();
    test_scoring_units();
    test_ion_series();
    test_heap();
    test_spectrum();
    test_enzyme();
    test_peptide();
    test_database();
    std::printf("oracle selftest: %d checks, %d failures\n", checks, failures);
    return failures == 0 ? 0 : 1;
}
