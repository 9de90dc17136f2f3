// ORACLE — TEST INFRASTRUCTURE ONLY (never linked into, imported by or executed from the product path).
//
// CPU restatement of the reference's post-search rescoring: the step that consumes the Feature
// records Scorer::score emits.  Sequential, f64 where the reference is f64, f32 where it is f32, -ffp-contract=off.
//   crates/sage/src/ml/matrix.rs, gauss.rs:27-165         Matrix / Gauss-Jordan solve with the eps ladder
//   crates/sage/src/ml/mod.rs:22-34                       mean / std
//   crates/sage/src/ml/kde.rs:14-169                      Kde, Builder::build, Estimator::posterior_error
//   crates/sage/src/ml/linear_discriminant.rs:57-231      LinearDiscriminantAnalysis::train / score, score_psms
//   crates/sage/src/ml/qvalue.rs:8-36                     spectrum_q_value
//   crates/sage/src/fdr.rs:42-226                         Competition::assign_q_value, picked_peptide, picked_protein
//   crates/sage-cli/src/runner.rs:281-292                 spectrum_fdr (heuristic fall-back, sort, q-values)
//   crates/sage-cli/src/runner.rs:513-530                 the predict_rt block: poisson-sorted q-values, then
//   crates/sage/src/ml/retention_alignment.rs:26-173      global_alignment
//   crates/sage/src/ml/regression.rs:21-122               LinearRegression::fit (streaming OLS, r^2)
//   crates/sage/src/ml/retention_model.rs:14-90           RetentionModel::embed / fit / predict
//   crates/sage/src/ml/mobility_model.rs:14-186           MobilityModel::embed / fit / predict
// Pinned by the known-answer tests the reference holds for these steps: linear_discriminant.rs:238-288 (LDA on 8 rows,
// normalised scores to 1e-8), regression.rs:124-157 (perfect line, noisy line, empty filter), mobility_model.rs:188-267
// (terminal-residue embedding counts) — tests/test_rescore_oracle.py.  KDE, q-values and the picked competitions have no reference
// vectors: for those this restatement IS the reference ("parity thinly pinned").
//
// Freedoms the reference leaves open, fixed here (and in the product) so that results are reproducible:
//   * rayon fold/sum order inside Kde::pdf (kde.rs:38-46) — here: sample order;
//   * par_sort_unstable_by on the discriminant score (runner.rs:290) and the order rows reach the stable par_sort_by of
//     fdr.rs:86 (hash-map iteration order) — here: stable sorts, rows in (key ascending, forward before reverse) order.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <numeric>
#include <vector>

// The arithmetic contract the PRODUCT fixes for this step (deterministic ln_1p / exp, blocked order of the KDE sums), in the
// checker's own copy: pure arithmetic, pinned against the platform libm and against the product's header in
// oracle/selftest.cpp; the reference-order mode below does not use it.
#include "detmath_oracle.h"
namespace sagedet = orcdet;

namespace {

// Two evaluation modes of the same restatement:
//   det == 0  the reference's own order: every sum strictly left to right (`iter().sum()`), ln_1p / exp from the platform
//             libm (what Rust's f64::ln_1p / exp call).  Kde::pdf, whose order the reference leaves to rayon, in sample order.
//   det == 1  the order and elementary functions the device evaluates: KDE / bandwidth sums in blocks of DET_BLOCK (identical to
//             det == 0 for n <= DET_BLOCK), ln_1p / exp from IEEE basic operations.  The device is held to this bit for bit.
//   The LDA sums (lda_train) are strictly sequential in both modes — and on the device.
thread_local int g_det = 0;

double ln1p(double x) { return g_det ? sagedet::det_log1p(x) : std::log1p(x); }
float ln1pf(float x) { return g_det ? sagedet::det_log1pf(x) : std::log1p(x); }
double expd(double x) { return g_det ? sagedet::det_exp(x) : std::exp(x); }
template <class Term>
double sum_terms(size_t n, Term term) {
    if (g_det) return sagedet::blocked_sum((uint64_t)n, term);
    double s = 0.0;
    for (size_t i = 0; i < n; ++i) s += term(i);
    return s;
}

struct Matrix {  // ml/matrix.rs: row-major f64
    std::vector<double> data;
    size_t rows = 0, cols = 0;
    Matrix() = default;
    Matrix(size_t r, size_t c) : data(r * c, 0.0), rows(r), cols(c) {}
    double& at(size_t i, size_t j) { return data[i * cols + j]; }
    double at(size_t i, size_t j) const { return data[i * cols + j]; }
    void swap_rows(size_t i, size_t j) {  // gauss.rs:17-25
        for (size_t k = 0; k < cols; ++k) std::swap(at(i, k), at(j, k));
    }
};

struct Gauss {  // gauss.rs:11-165
    Matrix left, right;

    void fill_zero(double eps) {  // :62-66
        for (size_t i = 0; i < left.cols; ++i) left.at(i, i) += eps;
    }
    bool left_solved() const {  // :69-87 (no abs on the off-diagonal test — restated as written)
        size_t n = left.cols;
        for (size_t i = 0; i < n; ++i)
            for (size_t j = 0; j < n; ++j) {
                double x = left.at(i, j);
                if (i == j) {
                    if (x != 1.0 && x != 0.0) return false;
                } else if (x > 1e-8) {
                    return false;
                }
            }
        return true;
    }
    void echelon() {  // :89-124
        size_t m = left.rows, n = left.cols, h = 0, k = 0;
        while (h < m && k < n) {
            size_t mi = 0;
            double mv = std::numeric_limits<double>::lowest();  // f64::MIN
            for (size_t i = h; i < m; ++i)
                if (left.at(i, k) >= mv) {
                    mi = i;
                    mv = left.at(i, k);
                }
            size_t i = mi;
            if (left.at(i, k) == 0.0) {
                ++k;
                continue;
            }
            if (h != mi) {
                left.swap_rows(h, i);
                right.swap_rows(h, i);
            }
            for (size_t r = h + 1; r < m; ++r) {
                double factor = left.at(r, k) / left.at(h, k);
                left.at(r, k) = 0.0;
                for (size_t j = k + 1; j < n; ++j) left.at(r, j) -= left.at(h, j) * factor;
                for (size_t j = 0; j < right.cols; ++j) right.at(r, j) -= right.at(h, j) * factor;
            }
            ++h;
            ++k;
        }
    }
    void reduce() {  // :127-143
        for (size_t ii = left.rows; ii-- > 0;)
            for (size_t j = 0; j < left.cols; ++j) {
                double x = left.at(ii, j);
                if (x == 0.0) continue;
                for (size_t k = j; k < left.cols; ++k) left.at(ii, k) /= x;
                for (size_t k = 0; k < right.cols; ++k) right.at(ii, k) /= x;
                break;
            }
    }
    void backfill() {  // :146-164
        for (size_t ii = left.rows; ii-- > 0;)
            for (size_t j = 0; j < left.cols; ++j) {
                if (left.at(ii, j) == 0.0) continue;
                for (size_t k = 0; k < ii; ++k) {
                    double factor = left.at(k, j) / left.at(ii, j);
                    for (size_t h = 0; h < left.cols; ++h) left.at(k, h) -= left.at(ii, h) * factor;
                    for (size_t h = 0; h < right.cols; ++h) right.at(k, h) -= right.at(ii, h) * factor;
                }
                break;
            }
    }
    static bool solve_inner(const Matrix& l, const Matrix& r, double eps, Matrix& out) {  // :28-41
        Gauss g{l, r};
        g.fill_zero(eps);
        g.echelon();
        g.reduce();
        g.backfill();
        if (!g.left_solved()) return false;
        out = g.right;
        return true;
    }
    static bool solve(const Matrix& l, const Matrix& r, Matrix& out) {  // :43-52
        double eps = 1e-8;
        while (eps <= 1.0) {
            if (solve_inner(l, r, eps, out)) return true;
            eps *= 10.0;
        }
        return false;
    }
};

double mean_of(const std::vector<double>& s) {  // ml/mod.rs:22-24
    double sum = sum_terms(s.size(), [&](size_t i) { return s[i]; });
    return sum / (double)s.size();
}
double std_of(const std::vector<double>& s) {  // ml/mod.rs:26-30
    double m = mean_of(s);
    double x = sum_terms(s.size(), [&](size_t i) { return (s[i] - m) * (s[i] - m); });
    return std::sqrt(x / (double)s.size());
}

size_t sat_usize(double x) {  // Rust `f64 as usize`: NaN -> 0, saturating
    if (!(x == x) || x <= 0.0) return 0;
    if (x >= 18446744073709551615.0) return std::numeric_limits<size_t>::max();
    return (size_t)x;
}

struct Kde {  // kde.rs:14-51
    const std::vector<double>* sample;
    double bandwidth, constant;
    Kde(const std::vector<double>& s, double bw_mult) : sample(&s) {
        double sigma = std_of(s);
        bandwidth = (sigma * std::pow((4.0 / 3.0) / (double)s.size(), 1.0 / 5.0)) * bw_mult;  // bw_adjust = x * bw_mult
        constant = std::sqrt(2.0 * M_PI) * bandwidth * (double)s.size();
    }
    double pdf(double x) const {
        const std::vector<double>& sm = *sample;
        double sum = sum_terms(sm.size(), [&](size_t i) {
            double u = (x - sm[i]) / bandwidth;
            return expd(-0.5 * (u * u));
        });
        return sum / constant;
    }
};

struct Estimator {  // kde.rs:136-169
    std::vector<double> bins;
    double min_score = 0, score_step = 0;
    double posterior_error(double score) const {
        size_t last = bins.size() ? bins.size() - 1 : 0;
        size_t lo = std::min(last, sat_usize(std::floor((score - min_score) / score_step)));
        size_t hi = std::min(last, lo + 1);
        double lower = bins[lo], upper = bins[hi];
        double lo_score = (double)lo * score_step + min_score;
        double linear = (score - lo_score) / score_step;
        return lower + ((upper - lower) * linear);
    }
};

Estimator kde_build(const std::vector<double>& scores, const std::vector<uint8_t>& decoys, bool monotonic, size_t nbins,
                    double bw_mult) {  // Builder::build, kde.rs:85-133
    std::vector<double> d, t;
    for (size_t i = 0; i < scores.size(); ++i) (decoys[i] ? d : t).push_back(scores[i]);
    double pi = (double)d.size() / (double)scores.size();
    Kde decoy(d, bw_mult), target(t, bw_mult);
    double mn = std::numeric_limits<double>::max(), mx = std::numeric_limits<double>::lowest();
    for (double s : scores) {  // f64::min / max ignore a NaN operand
        mn = std::fmin(mn, s);
        mx = std::fmax(mx, s);
    }
    Estimator e;
    e.min_score = mn;
    e.score_step = (mx - mn) / (double)(nbins - 1);
    e.bins.resize(nbins);
    for (size_t b = 0; b < nbins; ++b) {
        double score = ((double)b * e.score_step) + mn;
        double dd = decoy.pdf(score) * pi;
        double tt = target.pdf(score) * (1.0 - pi);
        e.bins[b] = dd / (tt + dd);
    }
    if (monotonic) {  // :120-126 (f64::max: a NaN operand yields the other one)
        double acc = e.bins.back();
        for (size_t b = nbins; b-- > 0;) {
            acc = std::fmax(acc, e.bins[b]);
            e.bins[b] = acc;
        }
    }
    return e;
}

// LinearDiscriminantAnalysis::train, linear_discriminant.rs:57-127.  rows: n x d row-major.  Both passes run strictly in row
// order, one running sum per accumulator, exactly as the reference's loops do — in both modes: these sums decide, through the
// pivot search of the elimination, whether the model is fitted at all, and the device evaluates them in this order too.
bool lda_train(const double* rows, size_t n, size_t d, const uint8_t* decoy, std::vector<double>& coef) {
    std::vector<double> class_sum[2] = {std::vector<double>(d, 0.0), std::vector<double>(d, 0.0)};
    size_t class_count[2] = {0, 0};
    for (size_t i = 0; i < n; ++i) {  // :70-82
        int cls = decoy[i] ? 0 : 1;
        for (size_t j = 0; j < d; ++j) class_sum[cls][j] += rows[i * d + j];
        class_count[cls]++;
    }
    if (class_count[0] == 0 || class_count[1] == 0) return false;
    std::vector<double> class_mean[2] = {std::vector<double>(d), std::vector<double>(d)};
    for (int c = 0; c < 2; ++c)
        for (size_t j = 0; j < d; ++j) class_mean[c][j] = class_sum[c][j] / (double)class_count[c];
    Matrix scatter[2] = {Matrix(d, d), Matrix(d, d)};
    std::vector<double> centered(d);
    for (size_t i = 0; i < n; ++i) {  // :92-103
        int cls = decoy[i] ? 0 : 1;
        for (size_t j = 0; j < d; ++j) centered[j] = rows[i * d + j] - class_mean[cls][j];
        for (size_t j = 0; j < d; ++j)
            for (size_t k = 0; k < d; ++k) scatter[cls].at(j, k) += centered[j] * centered[k];
    }
    Matrix within(d, d);
    for (int c = 0; c < 2; ++c)
        for (size_t j = 0; j < d * d; ++j) within.data[j] += scatter[c].data[j] / (double)class_count[c];
    Matrix mu(d, 1);
    for (size_t j = 0; j < d; ++j) mu.data[j] = class_mean[1][j] - class_mean[0][j];
    Matrix sol;
    if (!Gauss::solve(within, mu, sol)) return false;
    coef = sol.data;
    return true;
}

struct OrcFeature {  // == include/sage_hip.h SageFeature
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
static_assert(sizeof(OrcFeature) == 120, "layout");

constexpr size_t FEATURES = 20;

double clamp(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

// Competition::assign_q_value (fdr.rs:60-120) over groups keyed 0..n_keys; key 0xFFFFFFFF = feature takes no part.
size_t picked(const uint32_t* key, uint32_t n_keys, const uint8_t* decoy, const float* score, size_t n, float threshold,
              float* q_out /* [n], untouched for key == ~0 */) {
    const float FMIN = std::numeric_limits<float>::lowest();
    std::vector<float> fwd(n_keys, FMIN), rev(n_keys, FMIN);
    std::vector<uint8_t> has_f(n_keys, 0), has_r(n_keys, 0), seen(n_keys, 0);
    for (size_t i = 0; i < n; ++i) {
        if (key[i] == 0xFFFFFFFFu) continue;
        uint32_t g = key[i];
        seen[g] = 1;
        if (decoy[i]) {  // f32::max ignores a NaN operand
            rev[g] = std::fmax(rev[g], score[i]);
            has_r[g] = 1;
        } else {
            fwd[g] = std::fmax(fwd[g], score[i]);
            has_f[g] = 1;
        }
    }
    std::vector<double> ws;
    std::vector<uint8_t> wd;
    for (uint32_t g = 0; g < n_keys; ++g)
        if (seen[g]) {
            ws.push_back((double)std::fmax(fwd[g], rev[g]));  // score(), fdr.rs:43-45
            wd.push_back(rev[g] >= fwd[g]);                    // is_decoy(), :47-49
        }
    if (ws.empty()) return 0;
    Estimator est = kde_build(ws, wd, true, 1000, 1.0);
    struct Row {
        uint32_t g;
        uint8_t decoy;
        float score, q;
    };
    std::vector<Row> rows;
    for (uint32_t g = 0; g < n_keys; ++g) {
        if (has_f[g]) rows.push_back({g, 0, fwd[g], 1.0f});
        if (has_r[g]) rows.push_back({g, 1, rev[g], 1.0f});
    }
    auto total_key = [](float f) {  // f32::total_cmp
        int32_t b;
        std::memcpy(&b, &f, 4);
        return b ^ (int32_t)((uint32_t)(b >> 31) >> 1);
    };
    std::stable_sort(rows.begin(), rows.end(), [&](const Row& a, const Row& b) { return total_key(a.score) > total_key(b.score); });
    float dsum = 1.0f, tsum = 0.0f;
    for (auto& r : rows) {
        float pep = (float)est.posterior_error((double)r.score);
        dsum += pep;
        if (!r.decoy) tsum += 1.0f;
        r.q = dsum / tsum;
    }
    float q_min = 1.0f;
    size_t passing = 0;
    std::vector<float> qf(n_keys, 1.0f), qr(n_keys, 1.0f);
    for (size_t i = rows.size(); i-- > 0;) {
        q_min = std::fmin(q_min, rows[i].q);
        rows[i].q = q_min;
        if (q_min <= threshold && !rows[i].decoy) passing++;
        (rows[i].decoy ? qr : qf)[rows[i].g] = q_min;
    }
    for (size_t i = 0; i < n; ++i)
        if (key[i] != 0xFFFFFFFFu) q_out[i] = decoy[i] ? qr[key[i]] : qf[key[i]];
    return passing;
}

}  // namespace

extern "C" {

// LDA on an explicit n x d design (the reference's known-answer test drives train() this way). 1 = fitted.
// 0: the reference's order + platform libm (default); 1: the device's arithmetic contract (detmath.h).  Per calling thread.
void orc_rescore_mode(int det) { g_det = det ? 1 : 0; }

int orc_lda_train(const double* rows, uint64_t n, uint64_t d, const uint8_t* decoy, double* coef) {
    std::vector<double> c;
    if (!lda_train(rows, n, d, decoy, c)) return 0;
    std::memcpy(coef, c.data(), d * sizeof(double));
    return 1;
}

int orc_gauss_solve(const double* left, const double* right, uint64_t n, double* out) {
    Matrix l(n, n), r(n, 1), s;
    std::memcpy(l.data.data(), left, n * n * 8);
    std::memcpy(r.data.data(), right, n * 8);
    if (!Gauss::solve(l, r, s)) return 0;
    std::memcpy(out, s.data.data(), n * 8);
    return 1;
}

// kde::Builder{monotonic, bins, bw_adjust = x * bw_mult}.build(scores, decoys): bins + (min_score, score_step), then
// Estimator::posterior_error at `nq` query points.
void orc_kde(const double* scores, const uint8_t* decoys, uint64_t n, int monotonic, uint64_t nbins, double bw_mult,
             double* out_bins, double* out_min_step, const double* queries, uint64_t nq, double* out_pep) {
    std::vector<double> s(scores, scores + n);
    std::vector<uint8_t> d(decoys, decoys + n);
    Estimator e = kde_build(s, d, monotonic != 0, nbins, bw_mult);
    if (out_bins) std::memcpy(out_bins, e.bins.data(), nbins * 8);
    if (out_min_step) {
        out_min_step[0] = e.min_score;
        out_min_step[1] = e.score_step;
    }
    for (uint64_t i = 0; i < nq; ++i) out_pep[i] = e.posterior_error(queries[i]);
}

// runner.rs:536-541 on `n` Features: spectrum_fdr (score_psms or the heuristic, sort, spectrum_q_value), picked_peptide,
// picked_protein.  Outputs in INPUT order plus `order` = the permutation the reference leaves `features` in.
// tol_kind: 0 ppm, 2 da.  aligned_rt / delta_rt_model / delta_ims_model may be NULL (defaults of scoring.rs:576-592).
// rows_out (optional, n x 20) receives the LDA design.  Returns 1 when the linear model was fitted, 0 on the fall-back.
int orc_rescore(const OrcFeature* f, uint64_t n, int tol_kind, float tol_lo, float tol_hi, const float* aligned_rt,
                const float* delta_rt_model, const float* delta_ims_model, const uint32_t* peptide_key,
                uint32_t n_peptide_keys, const uint32_t* protein_key, uint32_t n_protein_keys, float* discriminant,
                float* posterior_error, float* spectrum_q, float* peptide_q, float* protein_q, uint32_t* order,
                uint64_t* passing /* [3] */, double* coef_out /* [20] */, double* rows_out) {
    std::vector<uint8_t> decoys(n);
    for (uint64_t i = 0; i < n; ++i) decoys[i] = f[i].label == -1;
    auto mass_error = [&](const OrcFeature& x) -> double {  // linear_discriminant.rs:140-144
        return tol_kind == 0 ? (double)x.delta_mass : (double)(x.expmass - x.calcmass);
    };
    double bw_adjust = tol_kind == 0 ? 2.0 : 0.1;  // :146-150
    float bin_size = tol_kind == 0 ? std::fmax(tol_hi - tol_lo, 100.0f) : std::fmax(tol_hi - tol_lo, 1000.0f);
    std::vector<double> delta_mass(n);
    for (uint64_t i = 0; i < n; ++i) delta_mass[i] = mass_error(f[i]);
    bool fitted = false;
    std::vector<double> coef;
    std::vector<double> rows(n * FEATURES);
    if (n) {
        Estimator mass_model = kde_build(delta_mass, decoys, false, (size_t)std::fabs(std::ceil(bin_size)), bw_adjust);
        for (uint64_t i = 0; i < n; ++i) {  // compute_features, :162-195
            const OrcFeature& p = f[i];
            double poisson = ln1p(-p.poisson);
            if (!std::isfinite(poisson)) poisson = 3.5;
            double* r = &rows[i * FEATURES];
            r[0] = (double)p.rank;
            r[1] = (double)p.charge;
            r[2] = ln1p(p.hyperscore);
            r[3] = ln1p(p.delta_next);
            r[4] = ln1p(p.delta_best);
            r[5] = mass_model.posterior_error(mass_error(p));
            r[6] = (double)p.isotope_error;
            r[7] = (double)p.average_ppm;
            r[8] = poisson;
            r[9] = ln1p((double)p.matched_intensity_pct);
            r[10] = (double)p.matched_peaks;
            r[11] = ln1p((double)p.longest_b);
            r[12] = ln1p((double)p.longest_y);
            r[13] = (double)p.longest_y / (double)p.peptide_len;
            r[14] = ln1p((double)p.peptide_len);
            r[15] = (double)p.missed_cleavages;
            r[16] = (double)(aligned_rt ? aligned_rt[i] : p.rt);
            r[17] = (double)p.ims;
            r[18] = std::sqrt(clamp((double)(delta_rt_model ? delta_rt_model[i] : 0.999f), 0.001, 0.999));
            r[19] = std::sqrt(clamp((double)(delta_ims_model ? delta_ims_model[i] : 0.999f), 0.001, 0.999));
        }
        if (rows_out) std::memcpy(rows_out, rows.data(), rows.size() * 8);
        fitted = lda_train(rows.data(), n, FEATURES, decoys.data(), coef);
        if (fitted)
            for (double c : coef)
                if (!std::isfinite(c)) fitted = false;  // :198-210
    }
    for (uint64_t i = 0; i < n; ++i) posterior_error[i] = 1.0f;  // Feature default, scoring.rs:580
    if (fitted) {
        std::vector<double> disc(n);
        for (uint64_t i = 0; i < n; ++i) {  // lda.score, :130-133 (iterator sum from 0.0)
            double s = 0.0;
            for (size_t j = 0; j < FEATURES; ++j) s += coef[j] * rows[i * FEATURES + j];
            disc[i] = s;
        }
        Estimator kde = kde_build(disc, decoys, true, 1000, 1.0);
        for (uint64_t i = 0; i < n; ++i) {  // :219-229
            discriminant[i] = (float)disc[i];
            float pe = (float)std::log10(kde.posterior_error(disc[i]));
            if (std::isinf(pe)) pe = -324.0f;
            posterior_error[i] = pe;
        }
        if (coef_out) std::memcpy(coef_out, coef.data(), FEATURES * 8);
    } else {
        for (uint64_t i = 0; i < n; ++i)  // runner.rs:285-288
            discriminant[i] = ln1pf((float)(-f[i].poisson)) + f[i].longest_y_pct / 3.0f;
    }
    // runner.rs:290 sort by discriminant descending (total_cmp), then qvalue.rs:8-36
    auto total_key = [](float x) {
        int32_t b;
        std::memcpy(&b, &x, 4);
        return b ^ (int32_t)((uint32_t)(b >> 31) >> 1);
    };
    std::vector<uint32_t> ord(n);
    std::iota(ord.begin(), ord.end(), 0u);
    std::stable_sort(ord.begin(), ord.end(),
                     [&](uint32_t a, uint32_t b) { return total_key(discriminant[a]) > total_key(discriminant[b]); });
    uint64_t dcount = 1, tcount = 0;
    for (uint32_t i : ord) {
        if (decoys[i]) dcount++;
        else tcount++;
        spectrum_q[i] = (float)dcount / (float)tcount;
    }
    float q_min = 1.0f;
    uint64_t pass = 0;
    for (uint64_t j = n; j-- > 0;) {
        uint32_t i = ord[j];
        q_min = std::fmin(q_min, spectrum_q[i]);
        spectrum_q[i] = q_min;
        if (q_min <= 0.01f) pass++;
    }
    if (order) std::memcpy(order, ord.data(), n * 4);
    passing[0] = pass;
    // picked competitions run over the features in their sorted order (fdr.rs:123-150, :152-187); the per-key maxima and
    // the q-value each (key, side) receives do not depend on that order.
    for (uint64_t i = 0; i < n; ++i) peptide_q[i] = protein_q[i] = 1.0f;
    passing[1] = picked(peptide_key, n_peptide_keys, decoys.data(), discriminant, n, 0.01f, peptide_q);
    passing[2] = picked(protein_key, n_protein_keys, decoys.data(), discriminant, n, 0.01f, protein_q);
    return fitted ? 1 : 0;
}

}  // extern "C"

// ---- the predict_rt block (runner.rs:513-530) --------------------------------------------------------------------------------
// LinearRegression::fit over the rows with filter[i] != 0 (regression.rs:58-122).  rows: n x d.  Returns false when no row
// passes or X^T X is singular.
static bool linreg_fit(const double* rows, const double* y, const uint8_t* filter, size_t n, size_t d, std::vector<double>& beta,
                       double& r2) {
    std::vector<double> cov(d * d, 0.0), b(d, 0.0);
    double sum_y = 0.0, sum_y2 = 0.0;
    size_t cnt = 0;
    for (size_t i = 0; i < n; ++i) {
        if (!filter[i]) continue;
        const double* row = rows + i * d;
        for (size_t j = 0; j < d; ++j) {  // Acc::add_row, :37-50
            const double rj = row[j];
            b[j] += rj * y[i];
            for (size_t k = 0; k < d; ++k) cov[j * d + k] += rj * row[k];
        }
        sum_y += y[i];
        sum_y2 += y[i] * y[i];
        cnt++;
    }
    if (cnt == 0) return false;
    const double nf = (double)cnt, y_mean = sum_y / nf, y_var = sum_y2 - nf * y_mean * y_mean;
    Matrix c(d, d), bm(d, 1), sol;
    c.data = cov;
    bm.data = b;
    if (!Gauss::solve(c, bm, sol)) return false;
    beta = sol.data;
    double sse = 0.0;
    for (size_t i = 0; i < n; ++i) {
        if (!filter[i]) continue;
        double pred = 0.0;
        for (size_t j = 0; j < d; ++j) pred += rows[i * d + j] * beta[j];
        sse += (pred - y[i]) * (pred - y[i]);
    }
    r2 = 1.0 - sse / y_var;
    return true;
}

static const uint8_t VALID_AA[22] = {'A', 'C', 'D', 'E', 'F', 'G', 'H', 'I', 'K', 'L', 'M',
                                     'N', 'P', 'Q', 'R', 'S', 'T', 'V', 'W', 'Y', 'U', 'O'};  // mass.rs:59-62
static void aa_map(size_t map[26]) {  // retention_model.rs:65-68
    for (int i = 0; i < 26; ++i) map[i] = 0;
    for (size_t i = 0; i < 22; ++i) map[VALID_AA[i] - 'A'] = i;
}
constexpr size_t RT_FEATURES = 22 * 3 + 3;  // retention_model.rs:33
constexpr size_t IM_FEATURES = 22 * 4 + 12;  // mobility_model.rs:78

// RetentionModel::embed (retention_model.rs:44-62)
static void rt_embed(const uint8_t* seq, size_t len, float mono, const size_t map[26], double* e) {
    for (size_t j = 0; j < RT_FEATURES; ++j) e[j] = 0.0;
    const size_t cterm = len >= 3 ? len - 3 : 0;
    for (size_t a = 0; a < len; ++a) {
        const size_t idx = map[seq[a] - 'A'];
        e[idx] += 1.0;
        if (a == 0 || a == 1) e[22 + idx] += 1.0;
        else if (a == cterm || a == cterm + 1) e[44 + idx] += 1.0;
    }
    e[RT_FEATURES - 3] = (double)len;
    e[RT_FEATURES - 2] = std::log1p((double)mono);
    e[RT_FEATURES - 1] = 1.0;
}

// MobilityModel::embed (mobility_model.rs:103-158).  The residue-class tables hold letter offsets (b'L' - b'A' ...) but are
// tested against the VALID_AA index of the residue (:121-139) — restated as written.
static bool in_set(size_t x, const size_t* set, size_t n) {
    for (size_t i = 0; i < n; ++i)
        if (set[i] == x) return true;
    return false;
}
static void im_embed(const uint8_t* seq, size_t len, float mono, uint8_t charge, const size_t map[26], double* e) {
    static const size_t BULKY[6] = {'L' - 'A', 'V' - 'A', 'I' - 'A', 'F' - 'A', 'W' - 'A', 'Y' - 'A'};
    static const size_t UC_POLAR[4] = {'S' - 'A', 'T' - 'A', 'N' - 'A', 'Q' - 'A'};
    static const size_t POSITIVE[3] = {'R' - 'A', 'K' - 'A', 'H' - 'A'};
    static const size_t NEGATIVE[2] = {'D' - 'A', 'E' - 'A'};
    static const size_t TINY[3] = {'G' - 'A', 'A' - 'A', 'S' - 'A'};
    static const size_t BRANCHED[3] = {'L' - 'A', 'I' - 'A', 'V' - 'A'};
    const size_t F = IM_FEATURES;
    for (size_t j = 0; j < F; ++j) e[j] = 0.0;
    const size_t cterm = len >= 3 ? len - 3 : 0;
    for (size_t a = 0; a < len; ++a) {
        const size_t idx = map[seq[a] - 'A'];
        e[idx] += 1.0;
        if (a == 0 || a == 1) e[44 + idx] += 1.0;  // N_TERMINAL = 22 * 2
        else if (a > cterm) e[66 + idx] += 1.0;     // C_TERMINAL = 22 * 3
        if (in_set(idx, BULKY, 6)) e[F - 9] += 1.0;
        if (in_set(idx, UC_POLAR, 4)) e[F - 10] += 1.0;
        if (in_set(idx, POSITIVE, 3)) e[F - 8] += 1.0;
        if (in_set(idx, NEGATIVE, 2)) e[F - 7] += 1.0;
        if (in_set(idx, TINY, 3)) e[F - 11] += 1.0;
        if (in_set(idx, BRANCHED, 3)) e[F - 12] += 1.0;
    }
    for (size_t i = 0; i < 22; ++i) e[22 + i] = e[i] / (double)len;  // PCT_FEATURES_START = 22
    const double z = (double)charge;
    e[F - 5] = z;                               // PEPTIDE_CHARGE
    e[F - 6] = 1.0 / z;                         // INV_PEPTIDE_CHARGE
    e[F - 3] = (double)len;                     // PEPTIDE_LEN
    e[F - 2] = (double)mono / 1000.0;           // PEPTIDE_MASS
    e[F - 4] = ((double)mono / z) / 1000.0;     // PEPTIDE_MZ
    e[F - 1] = 1.0;                             // INTERCEPT
}

extern "C" {

// LinearRegression::fit on an explicit design (the reference's unit tests drive it this way). 1 = fitted.
int orc_linreg_fit(const double* rows, const double* y, const uint8_t* filter, uint64_t n, uint64_t d, double* beta, double* r2) {
    std::vector<double> bb;
    double r = 0.0;
    if (!linreg_fit(rows, y, filter, n, d, bb, r)) return 0;
    std::memcpy(beta, bb.data(), d * 8);
    *r2 = r;
    return 1;
}

void orc_rt_embed(const uint8_t* seq, uint64_t len, float mono, double* out) {
    size_t map[26];
    aa_map(map);
    rt_embed(seq, len, mono, map, out);
}
void orc_im_embed(const uint8_t* seq, uint64_t len, float mono, uint8_t charge, double* out) {
    size_t map[26];
    aa_map(map);
    im_embed(seq, len, mono, charge, map, out);
}

// runner.rs:513-530 on `n` Features: sort by poisson + spectrum_q_value, global_alignment, retention_model::predict,
// mobility_model::predict.  seq_off / seq / mono: Peptide.sequence and .monoisotopic of db[f[i].peptide_idx], per Feature.
// alignments: [n_files][3] = {max_rt, slope, intercept}.  fitted / r2: [0] retention, [1] mobility.
void orc_predict_rt(const OrcFeature* f, uint64_t n, uint32_t n_files, const uint64_t* seq_off, const uint8_t* seq,
                    const float* mono, float* spectrum_q, float* aligned_rt, float* predicted_rt, float* delta_rt_model,
                    float* predicted_ims, float* delta_ims_model, float* alignments, int32_t* fitted, double* r2) {
    // ---- runner.rs:517-520: features.par_sort_unstable_by(poisson), spectrum_q_value ----
    auto total_key64 = [](double x) {
        int64_t b;
        std::memcpy(&b, &x, 8);
        return b ^ (int64_t)((uint64_t)(b >> 63) >> 1);
    };
    std::vector<uint32_t> ord(n);
    std::iota(ord.begin(), ord.end(), 0u);
    std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return total_key64(f[a].poisson) < total_key64(f[b].poisson); });
    {
        uint64_t d = 1, t = 0;
        for (uint32_t i : ord) {
            if (f[i].label == -1) d++;
            else t++;
            spectrum_q[i] = (float)d / (float)t;
        }
        float q_min = 1.0f;
        for (uint64_t j = n; j-- > 0;) {
            q_min = std::fmin(q_min, spectrum_q[ord[j]]);
            spectrum_q[ord[j]] = q_min;
        }
    }
    std::vector<uint8_t> train(n);
    for (uint64_t i = 0; i < n; ++i) train[i] = f[i].label == 1 && spectrum_q[i] <= 0.01f;

    // ---- global_alignment (retention_alignment.rs:100-173) ----
    std::vector<double> max_rt(n_files, 0.0);  // max_rt_by_file, :26-41 (ceil as u32)
    {
        std::vector<uint32_t> m(n_files, 0);
        for (uint64_t i = 0; i < n; ++i) {
            const float c = std::ceil(f[i].rt);
            const uint32_t v = !(c == c) || c <= 0.0f ? 0u : (c >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)c);  // `as u32`
            m[f[i].file_id] = std::max(m[f[i].file_id], v);
        }
        for (uint32_t k = 0; k < n_files; ++k) max_rt[k] = (double)m[k];
    }
    // mean_rt_by_file (:45-60): per (peptide, file) the MINIMUM rt of the training PSMs; rows in ascending peptide order
    std::vector<std::pair<uint32_t, uint64_t>> tr;
    for (uint64_t i = 0; i < n; ++i)
        if (train[i]) tr.push_back({f[i].peptide_idx, i});
    std::stable_sort(tr.begin(), tr.end(), [](auto& a, auto& b) { return a.first < b.first; });
    std::vector<std::vector<double>> mat;  // rt_matrix (:62-90): rows with a normal mean
    for (size_t a = 0; a < tr.size();) {
        size_t b = a;
        std::vector<double> v(n_files, std::nan(""));
        while (b < tr.size() && tr[b].first == tr[a].first) {
            const OrcFeature& x = f[tr[b].second];
            const double rt = (double)x.rt;
            v[x.file_id] = v[x.file_id] == v[x.file_id] ? std::fmin(v[x.file_id], rt) : rt;
            ++b;
        }
        double sum = 0.0, len = 0.0;
        for (uint32_t k = 0; k < n_files; ++k)
            if (v[k] == v[k]) {
                v[k] = v[k] / max_rt[k];
                sum += v[k];
                len += 1.0;
            }
        if (std::isnormal(sum / len)) mat.push_back(v);
        a = b;
    }
    std::vector<double> mean_rts(mat.size());  // :104-115
    for (size_t r = 0; r < mat.size(); ++r) {
        size_t len = 0;
        double sum = 0.0;
        for (uint32_t k = 0; k < n_files; ++k)
            if (std::isfinite(mat[r][k])) {
                len++;
                sum += mat[r][k];
            }
        mean_rts[r] = sum / (double)len;
    }
    std::vector<float> slope_f(n_files), icpt_f(n_files);
    for (uint32_t k = 0; k < n_files; ++k) {  // :118-163
        size_t len = 0;
        double dot = 0.0, sum_x = 0.0, sum_y = 0.0;
        for (size_t r = 0; r < mat.size(); ++r) {
            const double x = mat[r][k];
            if (!std::isfinite(x)) continue;
            len++;
            dot += x * mean_rts[r];
            sum_x += x;
            sum_y += mean_rts[r];
        }
        const double x_mean = sum_x / (double)len, y_mean = sum_y / (double)len;
        const double ssxy = dot - (double)len * x_mean * y_mean;
        double sx2 = 1e-8;
        for (size_t r = 0; r < mat.size(); ++r)
            if (std::isfinite(mat[r][k])) sx2 += (mat[r][k] - x_mean) * (mat[r][k] - x_mean);
        double slope = ssxy / sx2, intercept = y_mean - slope * x_mean;
        if (!std::isfinite(slope)) slope = 1.0;
        if (!std::isfinite(intercept)) intercept = 0.0;
        alignments[3 * k] = (float)max_rt[k];
        alignments[3 * k + 1] = slope_f[k] = (float)slope;
        alignments[3 * k + 2] = icpt_f[k] = (float)intercept;
    }
    for (uint64_t i = 0; i < n; ++i) {  // :165-172 (f32 arithmetic)
        const uint32_t k = f[i].file_id;
        aligned_rt[i] = (f[i].rt / (float)max_rt[k]) * slope_f[k] + icpt_f[k];
    }

    // ---- retention_model::predict (retention_model.rs:14-26), mobility_model::predict (mobility_model.rs:14-32) ----
    size_t map[26];
    aa_map(map);
    for (uint64_t i = 0; i < n; ++i) {  // Feature defaults, scoring.rs:576-592
        predicted_rt[i] = 0.0f;
        delta_rt_model[i] = 0.999f;
        predicted_ims[i] = 0.0f;
        delta_ims_model[i] = 0.999f;
    }
    fitted[0] = fitted[1] = 0;
    r2[0] = r2[1] = 0.0;
    {
        std::vector<double> rows(n * RT_FEATURES), y(n), beta;
        for (uint64_t i = 0; i < n; ++i) {
            rt_embed(seq + seq_off[i], seq_off[i + 1] - seq_off[i], mono[i], map, &rows[i * RT_FEATURES]);
            y[i] = (double)aligned_rt[i];
        }
        if (linreg_fit(rows.data(), y.data(), train.data(), n, RT_FEATURES, beta, r2[0])) {
            fitted[0] = 1;
            for (uint64_t i = 0; i < n; ++i) {
                double rt = 0.0;
                for (size_t j = 0; j < RT_FEATURES; ++j) rt = rt + rows[i * RT_FEATURES + j] * beta[j];
                const float bounded = (float)clamp(rt, 0.0, 1.0);
                predicted_rt[i] = bounded;
                delta_rt_model[i] = std::fabs(aligned_rt[i] - bounded);
            }
        }
    }
    {
        std::vector<double> rows(n * IM_FEATURES), y(n), beta;
        for (uint64_t i = 0; i < n; ++i) {
            im_embed(seq + seq_off[i], seq_off[i + 1] - seq_off[i], mono[i], f[i].charge, map, &rows[i * IM_FEATURES]);
            y[i] = (double)f[i].ims;
        }
        if (linreg_fit(rows.data(), y.data(), train.data(), n, IM_FEATURES, beta, r2[1])) {
            fitted[1] = 1;
            for (uint64_t i = 0; i < n; ++i) {
                double ims = 0.0;
                for (size_t j = 0; j < IM_FEATURES; ++j) ims = ims + rows[i * IM_FEATURES + j] * beta[j];
                const float bounded = (float)clamp(ims, 0.0, 2.0);
                predicted_ims[i] = bounded;
                delta_ims_model[i] = std::fabs(f[i].ims - bounded);
            }
        }
    }
}

}  // extern "C"
