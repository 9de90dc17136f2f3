// rescore.hip — post-search rescoring on the device: the consumer of the Feature records.
//
//   score_psms            crates/sage/src/ml/linear_discriminant.rs:133-231
//   LDA::train / score    linear_discriminant.rs:57-133        (two streaming passes: class means, within-class scatter)
//   Gauss::solve          ml/gauss.rs:27-165                   (host: 20x20)
//   kde::Builder::build   ml/kde.rs:85-133, Estimator::posterior_error :143-168
//   spectrum_fdr          sage-cli/src/runner.rs:281-292, ml/qvalue.rs:8-36
//   picked_peptide/protein, Competition::assign_q_value        fdr.rs:60-187
//
// What runs where.  Everything that is O(n) or O(n x bins) is a kernel: the 20-column design (f64, never leaves HBM), the
// per-class sums and the 2 x 20 x 20 scatter accumulation, the projection, the kernel-density sums (bins x samples Gaussian
// evaluations — the only compute-heavy part), the descending sorts (rocprim radix sort on the f32 total order), the
// decoy/target prefix counts and the suffix minimum of the q-values, the per-key maxima of the picked competitions.  The
// host does the scalar glue between kernels: final addition of per-block partials, bandwidths, the Gauss-Jordan solve.
//
// Every long f64 reduction (class sums, within-class scatter, KDE means / deviations / Gaussian sums) is evaluated in the
// BLOCKED ORDER of detmath.h — consecutive blocks of DET_BLOCK elements, left to right inside a block (one sequential chain
// per (block, accumulator) on the device), block sums added left to right — and ln_1p / exp come from detmath.h's IEEE-only
// implementations.  That fixes every bit that reaches the Gauss-Jordan pivot search, so the fit-or-heuristic decision
// (gauss.rs:69-124 compares with `>=` / `== 0.0`; constant columns such as ims == 0 make some pivots candidates pure
// rounding noise) and everything downstream equal the CPU restatement of the same contract bit for bit; for n <= DET_BLOCK
// the order is the reference's own sequential one.  The one sum the reference keeps in f32 — `decoy += pep` of
// fdr.rs:93-101 — is evaluated strictly in order (seq_cumsum_kernel).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/reverse_iterator.hpp>

#include "../../include/sage_hip.h"
#include "detmath.h"

namespace sagehip {

namespace {

constexpr int NF = 20;           // FEATURES, linear_discriminant.rs:19
constexpr int RB = 256;          // threads of a row-parallel block
constexpr int MAX_BLOCKS = 512;  // partials per (order-free) reduction
constexpr uint32_t DB = sagedet::DET_BLOCK;  // elements per block of the blocked summation order

struct KdeDev {  // kde::Estimator on the device
    const double* bins;
    double min_score, score_step;
    uint32_t nbins;
};

// Estimator::posterior_error, kde.rs:146-168 (`as usize` saturates, NaN -> 0)
__device__ inline double kde_posterior_error(const KdeDev& e, double score) {
    const uint32_t last = e.nbins ? e.nbins - 1 : 0;
    const double r = floor((score - e.min_score) / e.score_step);
    const uint32_t lo = (!(r == r) || r <= 0.0) ? 0u : (r >= (double)last ? last : (uint32_t)r);
    const uint32_t hi = lo + 1 < last ? lo + 1 : last;
    const double lower = e.bins[lo], upper = e.bins[hi];
    const double lo_score = (double)lo * e.score_step + e.min_score;
    const double linear = (score - lo_score) / e.score_step;
    return lower + ((upper - lower) * linear);
}

__device__ inline uint32_t total_order_key(float f) {  // ascending u32 order == f32::total_cmp
    const uint32_t b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ inline float from_total_order_key(uint32_t k) {
    return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

struct OpSum {
    __device__ double operator()(double a, double b) const { return a + b; }
};
struct OpMin {
    __device__ double operator()(double a, double b) const { return fmin(a, b); }
};
struct OpMax {
    __device__ double operator()(double a, double b) const { return fmax(a, b); }
};

// fixed-tree block reduction (wave shuffles, then the wave results through LDS); result valid in thread 0
template <class Op>
__device__ inline double block_reduce(double v, Op op, double* lds /* >= blockDim/64 doubles */) {
    for (int o = 32; o > 0; o >>= 1) v = op(v, __shfl_down(v, o, 64));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0)
        for (int w = 1; w < nw; ++w) v = op(v, lds[w]);
    return v;
}

// label -> decoy flag and the mass error of linear_discriminant.rs:140-144
__global__ __launch_bounds__(RB) void prep_kernel(const SageFeature* __restrict__ f, uint64_t n, int tol_kind,
                                                  uint8_t* __restrict__ decoy, double* __restrict__ dmass) {
    const uint64_t i = (uint64_t)blockIdx.x * RB + threadIdx.x;
    if (i >= n) return;
    decoy[i] = f[i].label == -1;
    dmass[i] = tol_kind == SAGE_TOL_PPM ? (double)f[i].delta_mass : (double)(f[i].expmass - f[i].calcmass);
}

// partial[b] = {min, max} over the block's slice (kde.rs:105-111; f64::min / max ignore a NaN operand, like fmin / fmax)
__global__ __launch_bounds__(RB) void minmax_kernel(const double* __restrict__ x, uint64_t n, double* __restrict__ partial) {
    __shared__ double lds[RB / 64];
    double mn = std::numeric_limits<double>::max(), mx = std::numeric_limits<double>::lowest();
    for (uint64_t i = (uint64_t)blockIdx.x * RB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * RB) {
        const double v = x[i];
        mn = fmin(mn, v);
        mx = fmax(mx, v);
    }
    const double r0 = block_reduce(mn, OpMin(), lds), r1 = block_reduce(mx, OpMax(), lds);
    if (threadIdx.x == 0) {
        partial[blockIdx.x * 2] = r0;
        partial[blockIdx.x * 2 + 1] = r1;
    }
}

// Builder::build's `d` / `t` vectors (kde.rs:86-98): the scores split by class, each class in input order.
// dpos[i] = number of decoys before i (exclusive scan of the flags); xs[0, nd) = decoys, xs[nd, n) = targets.
__global__ __launch_bounds__(RB) void class_flags_kernel(const uint8_t* __restrict__ decoy, uint64_t n, uint32_t* __restrict__ flags) {
    const uint64_t i = (uint64_t)blockIdx.x * RB + threadIdx.x;
    if (i < n) flags[i] = decoy[i] ? 1u : 0u;
}
__global__ __launch_bounds__(RB) void class_split_kernel(const double* __restrict__ x, const uint8_t* __restrict__ decoy,
                                                         const uint32_t* __restrict__ dpos, uint64_t n, uint64_t nd,
                                                         double* __restrict__ xs) {
    const uint64_t i = (uint64_t)blockIdx.x * RB + threadIdx.x;
    if (i >= n) return;
    const uint64_t d = dpos[i];
    xs[decoy[i] ? d : nd + (i - d)] = x[i];
}

// part[b] = left-to-right sum over block b of x[i] (mode 0: ml/mod.rs:22-24) or (x[i] - mean)^2 (mode 1: ml/mod.rs:26-30).
// One thread = one block = one sequential chain (the blocked order of detmath.h).
__global__ __launch_bounds__(64) void blocked_sum_kernel(const double* __restrict__ x, uint64_t n, int mode, double mean,
                                                         double* __restrict__ part) {
    const uint64_t b = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    const uint64_t lo = b * DB, hi = lo + DB < n ? lo + DB : n;
    if (lo >= n) return;
    double s = 0.0;
    for (uint64_t i = lo; i < hi; ++i) {
        const double v = x[i];
        s += mode ? (v - mean) * (v - mean) : v;
    }
    part[b] = s;
}

// Kde::pdf numerators (kde.rs:34-50) of ONE class: partial[blk * nbins + bin] = sum over sample block blk, left to right, of
// exp(-0.5 ((score_bin - x_i) / h)^2).  One thread = one (bin, block) chain; the block's samples are staged in LDS once and
// read with wave-uniform addresses.
constexpr int KDE_THREADS = 256;
__global__ __launch_bounds__(KDE_THREADS) void kde_pdf_kernel(const double* __restrict__ xs, uint64_t n_c, double min_score,
                                                              double score_step, uint32_t nbins, double h,
                                                              double* __restrict__ partial) {
    __shared__ double tile[DB];
    const uint64_t lo = (uint64_t)blockIdx.y * DB;
    const uint32_t cnt = (uint32_t)(n_c - lo < DB ? n_c - lo : DB);
    for (uint32_t i = threadIdx.x; i < cnt; i += KDE_THREADS) tile[i] = xs[lo + i];
    __syncthreads();
    const uint32_t bin = blockIdx.x * KDE_THREADS + threadIdx.x;
    if (bin >= nbins) return;
    const double centre = ((double)bin * score_step) + min_score;
    double acc = 0.0;
    for (uint32_t i = 0; i < cnt; ++i) {
        const double u = (centre - tile[i]) / h;
        acc += sagedet::det_exp(-0.5 * (u * u));
    }
    partial[(uint64_t)blockIdx.y * nbins + bin] = acc;
}

// bins[b] = decoy_pdf * pi / (target_pdf * (1 - pi) + decoy_pdf * pi), then the monotone fold (kde.rs:113-126).  The block
// partials of a bin are added left to right from 0.0 (second level of the blocked order).  One workgroup.
__global__ __launch_bounds__(1024) void kde_finish_kernel(const double* __restrict__ part_d, uint32_t nblk_d,
                                                          const double* __restrict__ part_t, uint32_t nblk_t, uint32_t nbins,
                                                          double const_d, double const_t, double pi, int monotonic,
                                                          double* __restrict__ bins) {
    for (uint32_t b = threadIdx.x; b < nbins; b += blockDim.x) {
        double sd = 0.0, st = 0.0;
        for (uint32_t s = 0; s < nblk_d; ++s) sd += part_d[(uint64_t)s * nbins + b];
        for (uint32_t s = 0; s < nblk_t; ++s) st += part_t[(uint64_t)s * nbins + b];
        const double d = (sd / const_d) * pi;
        const double t = (st / const_t) * (1.0 - pi);
        bins[b] = d / (t + d);
    }
    __syncthreads();
    if (monotonic && threadIdx.x == 0) {
        double acc = bins[nbins - 1];
        for (uint32_t b = nbins; b-- > 0;) {
            acc = fmax(acc, bins[b]);
            bins[b] = acc;
        }
    }
}

// compute_features, linear_discriminant.rs:162-195: one 20-column f64 row per Feature
__global__ __launch_bounds__(RB) void rows_kernel(const SageFeature* __restrict__ f, uint64_t n, const double* __restrict__ dmass,
                                                  KdeDev mass_model, const float* __restrict__ aligned_rt,
                                                  const float* __restrict__ delta_rt, const float* __restrict__ delta_ims,
                                                  double* __restrict__ rows) {
    const uint64_t i = (uint64_t)blockIdx.x * RB + threadIdx.x;
    if (i >= n) return;
    const SageFeature p = f[i];
    double poisson = sagedet::det_log1p(-p.poisson);
    if (!isfinite(poisson)) poisson = 3.5;
    double* r = rows + i * NF;
    r[0] = (double)p.rank;
    r[1] = (double)p.charge;
    r[2] = sagedet::det_log1p(p.hyperscore);
    r[3] = sagedet::det_log1p(p.delta_next);
    r[4] = sagedet::det_log1p(p.delta_best);
    r[5] = kde_posterior_error(mass_model, dmass[i]);
    r[6] = (double)p.isotope_error;
    r[7] = (double)p.average_ppm;
    r[8] = poisson;
    r[9] = sagedet::det_log1p((double)p.matched_intensity_pct);
    r[10] = (double)p.matched_peaks;
    r[11] = sagedet::det_log1p((double)p.longest_b);
    r[12] = sagedet::det_log1p((double)p.longest_y);
    r[13] = (double)p.longest_y / (double)p.peptide_len;
    r[14] = sagedet::det_log1p((double)p.peptide_len);
    r[15] = (double)p.missed_cleavages;
    r[16] = (double)(aligned_rt ? aligned_rt[i] : p.rt);
    r[17] = (double)p.ims;
    const double drt = (double)(delta_rt ? delta_rt[i] : 0.999f), dims = (double)(delta_ims ? delta_ims[i] : 0.999f);
    r[18] = sqrt(drt < 0.001 ? 0.001 : (drt > 0.999 ? 0.999 : drt));  // f64::clamp: NaN stays NaN
    r[19] = sqrt(dims < 0.001 ? 0.001 : (dims > 0.999 ? 0.999 : dims));
}

// train's two passes over the rows (linear_discriminant.rs:70-82 class sums, :92-103 within-class scatter) in the REFERENCE'S
// ORDER: every accumulator is one running f64 sum over the rows of its class, strictly in row order — these sums decide,
// through the signed-maximum pivot search of the elimination (gauss.rs:97-108), whether the model is fitted at all, so a
// blocked order is not good enough.  One lane per accumulator, one workgroup (8 wavefronts) per CLASS: 20 chains
// for the class sums (SCATTER == false), 400 matrix entries for the scatter.  The workgroup walks ALL rows, 128 at a time: its
// 512 threads stage the tile's rows OF THEIR CLASS into LDS, compacted in row order (two ballots give the class mask; for the
// scatter the rows are centred on the way in, `row[j] - mu[j]` as the reference forms it), so the chain of a lane is two LDS
// reads, a multiply and THE add per row, nothing else; the next tile's loads are in flight meanwhile.  A wavefront that runs
// nearly alone retires an instruction every ~6 cycles, so instructions per row are what this costs: ~25 ms per million rows
// for both passes (a blocked order took 2 ms; a lane-per-accumulator walk with a class test per row 110 ms).
constexpr int SEQ_TILE = 128;                                     // rows per staged tile
constexpr int SEQ_THREADS = 512;
constexpr int SEQ_PER_THREAD = SEQ_TILE * NF / SEQ_THREADS;       // elements of a tile a thread moves (5)
static_assert(SEQ_TILE * NF % SEQ_THREADS == 0 && SEQ_TILE == 128, "two 64-row class masks per tile");
template <bool SCATTER>
__global__ __launch_bounds__(SEQ_THREADS) void seq_lda_kernel(const double* __restrict__ rows, const uint8_t* __restrict__ decoy,
                                                              uint64_t n, const double* __restrict__ class_mean /* [2][20], SCATTER only */,
                                                              double* __restrict__ out /* [2][20] resp. [2][400] */) {
    __shared__ double stage[2][SEQ_TILE][NF];
    __shared__ unsigned long long cmask[2][2];  // [buffer][rows 0..63 | 64..127]: bit r = the row belongs to this workgroup's class
    __shared__ double mu[NF];
    const uint32_t t = threadIdx.x;
    const uint32_t cls = blockIdx.y;  // 0 = decoys, 1 = targets (linear_discriminant.rs:73)
    const bool owner = SCATTER ? t < NF * NF : t < NF;
    const uint32_t ej = owner ? (SCATTER ? t / NF : t) : 0, ek = owner && SCATTER ? t % NF : 0;
    if (SCATTER && t < NF) mu[t] = class_mean[cls * NF + t];
    __syncthreads();
    double reg[SEQ_PER_THREAD];
    bool reg_mine = false;
    auto load_tile = [&](uint64_t base) {  // element e = t + 512 i of the tile: row e / 20, column e % 20; thread t < 128: row t's class
        const uint64_t cnt = n - base < SEQ_TILE ? n - base : SEQ_TILE;
#pragma unroll
        for (int i = 0; i < SEQ_PER_THREAD; ++i) {
            const uint32_t e = t + (uint32_t)SEQ_THREADS * (uint32_t)i;
            reg[i] = e < cnt * NF ? rows[base * NF + e] : 0.0;
        }
        reg_mine = t < cnt && (decoy[base + t] ? 0u : 1u) == cls;
    };
    auto store_tile = [&](int buf) {  // (every thread calls it: there is a barrier inside)
        if (t < SEQ_TILE) {
            const unsigned long long m = __ballot(reg_mine);
            if ((t & 63u) == 0) cmask[buf][t >> 6] = m;
        }
        __syncthreads();
        const unsigned long long m0 = cmask[buf][0], m1 = cmask[buf][1];
#pragma unroll
        for (int i = 0; i < SEQ_PER_THREAD; ++i) {
            const uint32_t e = t + (uint32_t)SEQ_THREADS * (uint32_t)i, r = e / NF, c = e % NF;
            const unsigned long long half = r < 64 ? m0 : m1, bit = 1ull << (r & 63u);
            if (half & bit) {  // the row's place among the tile's rows of this class, in row order
                const uint32_t rank = (r < 64 ? 0u : (uint32_t)__popcll(m0)) + (uint32_t)__popcll(half & (bit - 1ull));
                stage[buf][rank][c] = SCATTER ? reg[i] - mu[c] : reg[i];
            }
        }
    };
    double acc = 0.0;
    if (n) load_tile(0);
    store_tile(0);
    __syncthreads();
    int buf = 0;
    for (uint64_t base = 0; base < n; base += SEQ_TILE, buf ^= 1) {
        const bool more = base + SEQ_TILE < n;
        if (more) load_tile(base + SEQ_TILE);
        if (owner) {
            const uint32_t cnt = (uint32_t)(__popcll(cmask[buf][0]) + __popcll(cmask[buf][1]));
            const double* pj = &stage[buf][0][ej];
            const double* pk = &stage[buf][0][ek];
#pragma unroll 4
            for (uint32_t r = 0; r < cnt; ++r) acc += SCATTER ? pj[r * NF] * pk[r * NF] : pj[r * NF];  // in row order
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
    }
    if (owner) out[(size_t)cls * (SCATTER ? NF * NF : NF) + t] = acc;
}

// out[c] = sum over blocks (in block order) of partial[b][c]
__global__ __launch_bounds__(RB) void fold_partials_kernel(const double* __restrict__ partial, uint32_t n_blocks, uint32_t width,
                                                           double* __restrict__ out) {
    const uint32_t c = blockIdx.x * RB + threadIdx.x;
    if (c >= width) return;
    double s = 0.0;
    for (uint32_t b = 0; b < n_blocks; ++b) s += partial[(uint64_t)b * width + c];
    out[c] = s;
}

struct Coef {
    double w[NF];
};

// lda.score (:130-133): left-to-right sum from 0.0, identical to the reference given identical rows and coefficients
__global__ __launch_bounds__(RB) void project_kernel(const double* __restrict__ rows, uint64_t n, Coef coef,
                                                     double* __restrict__ disc) {
    const uint64_t i = (uint64_t)blockIdx.x * RB + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < NF; ++j) s += coef.w[j] * rows[i * NF + j];
    disc[i] = s;
}

// :219-229
__global__ __launch_bounds__(RB) void pep_kernel(const double* __restrict__ disc, uint64_t n, KdeDev kde,
                                                 float* __restrict__ discriminant, float* __restrict__ posterior_error) {
    const uint64_t i = (uint64_t)blockIdx.x * RB + threadIdx.x;
    if (i >= n) return;
    discriminant[i] = (float)disc[i];
    float pe = (float)log10(kde_posterior_error(kde, disc[i]));
    if (isinf(pe)) pe = -324.0f;
    posterior_error[i] = pe;
}

// runner.rs:285-288 heuristic when the model cannot be fitted; posterior_error keeps the Feature default 1.0
__global__ __launch_bounds__(RB) void heuristic_kernel(const SageFeature* __restrict__ f, uint64_t n,
                                                       float* __restrict__ discriminant, float* __restrict__ posterior_error) {
    const uint64_t i = (uint64_t)blockIdx.x * RB + threadIdx.x;
    if (i >= n) return;
    discriminant[i] = sagedet::det_log1pf((float)(-f[i].poisson)) + f[i].longest_y_pct / 3.0f;
    posterior_error[i] = 1.0f;
}

__global__ __launch_bounds__(RB) void sort_keys_kernel(const float* __restrict__ score, uint32_t n, uint32_t* __restrict__ keys,
                                                       uint32_t* __restrict__ idx) {
    const uint32_t i = blockIdx.x * RB + threadIdx.x;
    if (i >= n) return;
    keys[i] = total_order_key(score[i]);
    idx[i] = i;
}

// flags[j] = 1 when the j-th best PSM is a decoy (input of the prefix count of qvalue.rs:16-24)
__global__ __launch_bounds__(RB) void decoy_flags_kernel(const uint32_t* __restrict__ order, const uint8_t* __restrict__ decoy,
                                                         uint32_t n, uint32_t* __restrict__ flags) {
    const uint32_t j = blockIdx.x * RB + threadIdx.x;
    if (j < n) flags[j] = decoy[order[j]] ? 1u : 0u;
}

// qvalue.rs:16-24 on the sorted order: q[j] = (1 + decoys in [0, j]) / (targets in [0, j]), already folded with the 1.0 the
// reverse cumulative minimum starts from (min is associative: min(1, min(a, b)) == min(min(1, a), min(1, b)))
__global__ __launch_bounds__(RB) void q_from_counts_kernel(const uint32_t* __restrict__ decoy_cum, uint32_t n, float* __restrict__ q) {
    const uint32_t j = blockIdx.x * RB + threadIdx.x;
    if (j >= n) return;
    const uint32_t d = 1u + decoy_cum[j], t = (j + 1u) - decoy_cum[j];  // `let mut decoy = 1; let mut target = 0;`
    q[j] = fminf(1.0f, (float)d / (float)t);
}

struct OpFmin {
    __device__ float operator()(float a, float b) const { return fminf(a, b); }
};

// rows passing `threshold` after the reverse cumulative minimum (qvalue.rs:31-33; fdr.rs:108-110 also requires a target row)
__global__ __launch_bounds__(RB) void count_passing_kernel(const float* __restrict__ q, uint32_t n, const uint8_t* __restrict__ row_decoy,
                                                           float threshold, unsigned long long* __restrict__ passing) {
    const uint32_t j = blockIdx.x * RB + threadIdx.x;
    const bool pass = j < n && q[j] <= threshold && !(row_decoy && row_decoy[j]);
    const unsigned long long b = __ballot(pass);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(passing, (unsigned long long)__popcll(b));
}

__global__ __launch_bounds__(RB) void scatter_by_order_kernel(const uint32_t* __restrict__ order, const float* __restrict__ q,
                                                              uint32_t n, float* __restrict__ out) {
    const uint32_t j = blockIdx.x * RB + threadIdx.x;
    if (j < n) out[order[j]] = q[j];
}

// ---- picked competitions (fdr.rs) --------------------------------------------------------------------------------------
// side[2 * key + s] (s = 0 forward, 1 reverse) holds the total-order key of the best discriminant of that side, 0 = absent
// (0 is the key of a negative NaN with full payload; every real score maps above it).
__global__ __launch_bounds__(RB) void picked_max_kernel(const uint32_t* __restrict__ key, const uint8_t* __restrict__ decoy,
                                                        const float* __restrict__ score, uint64_t n, uint32_t* __restrict__ side) {
    const uint64_t i = (uint64_t)blockIdx.x * RB + threadIdx.x;
    if (i >= n || key[i] == 0xFFFFFFFFu) return;
    float s = score[i];
    if (!(s == s)) s = -std::numeric_limits<float>::max();  // f32::max(f32::MIN, NaN) == f32::MIN (fdr.rs:137,141)
    s = fmaxf(s, -std::numeric_limits<float>::max());
    atomicMax(&side[2ull * key[i] + (decoy[i] ? 1 : 0)], total_order_key(s));
}

// per key: the winner of the competition for the KDE fit (Competition::score / is_decoy, fdr.rs:42-49) and the sort keys of
// its rows in (key, forward-before-reverse) order
__global__ __launch_bounds__(RB) void picked_rows_kernel(const uint32_t* __restrict__ side, uint32_t n_keys,
                                                         double* __restrict__ winner, uint8_t* __restrict__ winner_decoy,
                                                         uint32_t* __restrict__ row_key, uint32_t* __restrict__ row_id,
                                                         uint32_t* __restrict__ counters /* [0] rows present, [1] unused keys */) {
    const uint32_t g = blockIdx.x * RB + threadIdx.x;
    if (g >= n_keys) return;
    const uint32_t kf = side[2 * g], kr = side[2 * g + 1];
    const float FMIN = -std::numeric_limits<float>::max();
    const float fwd = kf ? from_total_order_key(kf) : FMIN, rev = kr ? from_total_order_key(kr) : FMIN;
    winner[g] = (double)fmaxf(fwd, rev);
    winner_decoy[g] = rev >= fwd;
    row_key[2 * g] = kf;
    row_key[2 * g + 1] = kr;
    row_id[2 * g] = 2 * g;
    row_id[2 * g + 1] = 2 * g + 1;
    const uint32_t present = (kf != 0) + (kr != 0);
    if (present) atomicAdd(&counters[0], present);
    else atomicAdd(&counters[1], 1u);
}

// rows in sorted order: their posterior error (fdr.rs:94), side, and the target flag the prefix count runs over; pep is
// zero-padded to a whole number of seq_cumsum_kernel tiles
__global__ __launch_bounds__(RB) void picked_pep_kernel(const uint32_t* __restrict__ sorted_key, const uint32_t* __restrict__ sorted_id,
                                                        uint32_t m, uint32_t m_padded, KdeDev est, float* __restrict__ pep,
                                                        uint8_t* __restrict__ row_decoy, uint32_t* __restrict__ target_flag) {
    const uint32_t j = blockIdx.x * RB + threadIdx.x;
    if (j >= m_padded) return;
    if (j >= m) {
        pep[j] = 0.0f;
        return;
    }
    pep[j] = (float)kde_posterior_error(est, (double)from_total_order_key(sorted_key[j]));
    const uint32_t dec = sorted_id[j] & 1;
    row_decoy[j] = (uint8_t)dec;
    target_flag[j] = dec ^ 1u;
}

// fdr.rs:91-101: decoy = 1.0; target = 0.0; for row { decoy += pep; if !row.decoy { target += 1.0 }; q = decoy / target }.
// `decoy` is an f32 running sum whose rounding depends on the order, so it is evaluated strictly in row order by ONE
// wavefront whose lanes all carry the same running sum: one dependent v_add_f32 per row.  The wavefront streams the pep
// values through LDS in tiles of 1024 rows (the next tile's global loads are in flight while the current one is summed),
// reads them back with uniform addresses, parks the running sums of 64 rows in LDS and lane k stores the sum as it stood
// after row k.  Nothing in the loop waits on global memory.  `target` counts whole numbers (exact in f32 below 2^24) and
// comes from the parallel prefix count; the division is picked_q_kernel's.
constexpr int CS_TILE = 1024;
__global__ __launch_bounds__(64) void seq_cumsum_kernel(const float* __restrict__ pep /* zero-padded to m_padded */,
                                                        uint32_t m_padded /* multiple of CS_TILE, > 0 */,
                                                        float* __restrict__ dsum) {
    __shared__ float4 tile[CS_TILE / 4];
    __shared__ float4 run[16];
    const uint32_t lane = threadIdx.x;
    const float4* src = reinterpret_cast<const float4*>(pep);
    float4 n0 = src[lane], n1 = src[64 + lane], n2 = src[128 + lane], n3 = src[192 + lane];
    float d = 1.0f;
    for (uint32_t base = 0; base < m_padded; base += CS_TILE) {
        tile[lane] = n0;
        tile[64 + lane] = n1;
        tile[128 + lane] = n2;
        tile[192 + lane] = n3;
        __syncthreads();
        {  // next tile (the last iteration re-reads its own tile: no branch, nothing spilled)
            const uint32_t nb = (base + CS_TILE < m_padded ? base + CS_TILE : base) / 4;
            n0 = src[nb + lane];
            n1 = src[nb + 64 + lane];
            n2 = src[nb + 128 + lane];
            n3 = src[nb + 192 + lane];
        }
        float4 cur[16], nxt[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) cur[i] = tile[i];  // uniform addresses: every lane reads the same 64 values
        for (int c = 0; c < CS_TILE / 64; ++c) {
            const int cn = c + 1 < CS_TILE / 64 ? c + 1 : c;
#pragma unroll
            for (int i = 0; i < 16; ++i) nxt[i] = tile[cn * 16 + i];  // in flight during the 64 dependent adds below
            float4 r[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                d = d + cur[i].x;
                r[i].x = d;
                d = d + cur[i].y;
                r[i].y = d;
                d = d + cur[i].z;
                r[i].z = d;
                d = d + cur[i].w;
                r[i].w = d;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) run[i] = r[i];
            __syncthreads();
            dsum[base + c * 64 + lane] = reinterpret_cast<const float*>(run)[lane];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 16; ++i) cur[i] = nxt[i];
        }
    }
}

// q = decoy / target, folded with the 1.0 the reverse minimum starts from (a NaN q — no target yet and a NaN posterior
// error — becomes 1.0, as f32::min makes it in fdr.rs:107)
__global__ __launch_bounds__(RB) void picked_q_kernel(const float* __restrict__ dsum, const uint32_t* __restrict__ target_cum,
                                                      uint32_t m, float* __restrict__ q) {
    const uint32_t j = blockIdx.x * RB + threadIdx.x;
    if (j < m) q[j] = fminf(1.0f, dsum[j] / (float)target_cum[j]);
}

__global__ __launch_bounds__(RB) void picked_scatter_kernel(const uint32_t* __restrict__ sorted_id, const float* __restrict__ q,
                                                            uint32_t m, float* __restrict__ side_q) {
    const uint32_t j = blockIdx.x * RB + threadIdx.x;
    if (j < m) side_q[sorted_id[j]] = q[j];
}

__global__ __launch_bounds__(RB) void picked_gather_kernel(const uint32_t* __restrict__ key, const uint8_t* __restrict__ decoy,
                                                           uint64_t n, const float* __restrict__ side_q, float* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * RB + threadIdx.x;
    if (i >= n) return;
    out[i] = key[i] == 0xFFFFFFFFu ? 1.0f : side_q[2ull * key[i] + (decoy[i] ? 1 : 0)];
}

// ---- host side ---------------------------------------------------------------------------------------------------------

// Device scratch of one call.  Allocations are stream-ordered (hipMallocAsync / hipFreeAsync on the call's stream) out of
// the device's default memory pool, whose release threshold is raised once so that freed blocks stay cached between
// calls: after the first call a rescoring allocates nothing from the driver (the ~30 hipMalloc / hipFree pairs used to be a
// third of the wall time of a 1 M-PSM call).  SAGE_HIP_NO_POOL=1 falls back to plain hipMalloc / hipFree.
thread_local hipStream_t tl_stream = nullptr;
thread_local bool tl_pooled = false;

void scratch_begin(int device, hipStream_t stream) {
    tl_stream = stream;
    tl_pooled = false;
    if (const char* e = getenv("SAGE_HIP_NO_POOL"))
        if (atoi(e) != 0) return;
    static bool configured[64] = {};
    hipMemPool_t pool = nullptr;
    if (hipDeviceGetDefaultMemPool(&pool, device) != hipSuccess || !pool) {
        (void)hipGetLastError();
        return;
    }
    if (device >= 0 && device < 64 && !configured[device]) {
        uint64_t keep = ~0ull;
        if (hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep) != hipSuccess) {
            (void)hipGetLastError();
            return;
        }
        configured[device] = true;
    }
    tl_pooled = true;
}

template <class T>
struct Buf {
    T* p = nullptr;
    bool pooled = false;
    hipStream_t stream = nullptr;
    Buf() = default;
    Buf(const Buf&) = delete;
    Buf& operator=(const Buf&) = delete;
    ~Buf() { release(); }
    void release() {
        if (!p) return;
        if (pooled) (void)hipFreeAsync(p, stream);  // ordered after every kernel of this call that uses it
        else (void)hipFree(p);
        p = nullptr;
    }
    hipError_t alloc(size_t count) {
        release();
        const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
        if (tl_pooled) {
            if (hipMallocAsync((void**)&p, bytes, tl_stream) == hipSuccess) {
                pooled = true;
                stream = tl_stream;
                return hipSuccess;
            }
            (void)hipGetLastError();
            p = nullptr;
        }
        pooled = false;
        return hipMalloc((void**)&p, bytes);
    }
};

struct EventPair {  // start / stop events of one call, destroyed on every exit path
    hipEvent_t start = nullptr, stop = nullptr;
    EventPair() = default;
    EventPair(const EventPair&) = delete;
    EventPair& operator=(const EventPair&) = delete;
    ~EventPair() {
        if (start) (void)hipEventDestroy(start);
        if (stop) (void)hipEventDestroy(stop);
    }
};

struct Ctx {
    hipStream_t stream = nullptr;
    std::string err;
    int code = SAGE_HIP_OK;
    bool check(hipError_t e, const char* what) {
        if (e == hipSuccess) return true;
        code = e == hipErrorOutOfMemory ? SAGE_HIP_ERR_OOM : SAGE_HIP_ERR_HIP;
        err = std::string(what) + ": " + hipGetErrorString(e);
        return false;
    }
};
#define RS_TRY(expr)                         \
    do {                                     \
        if (!cx.check((expr), #expr)) return false; \
    } while (0)

uint32_t grid_for(uint64_t n, uint32_t block) { return (uint32_t)std::max<uint64_t>(1, (n + block - 1) / block); }
uint32_t blocks_for(uint64_t n) { return (uint32_t)std::min<uint64_t>(MAX_BLOCKS, std::max<uint64_t>(1, (n + RB - 1) / RB)); }

struct KdeFit {
    Buf<double> bins;
    KdeDev dev{};
};

bool exclusive_count(Ctx& cx, const uint32_t* d_flags, uint32_t* d_out, uint64_t n);

// left-to-right fold of block partials from +0.0 (second level of the blocked order)
double fold_partials(const std::vector<double>& part) {
    double s = 0.0;
    for (double v : part) s += v;
    return s;
}

// blocked-order sum of x[0, n) (mode 0) or of (x - mean)^2 (mode 1): chains on the device, block partials folded on the host
bool blocked_sum(Ctx& cx, const double* d_x, uint64_t n, int mode, double mean, double& out) {
    out = 0.0;
    if (n == 0) return true;
    const uint64_t nblk = (n + DB - 1) / DB;
    Buf<double> part;
    RS_TRY(part.alloc(nblk));
    blocked_sum_kernel<<<grid_for(nblk, 64), 64, 0, cx.stream>>>(d_x, n, mode, mean, part.p);
    std::vector<double> h(nblk);
    RS_TRY(hipMemcpyAsync(h.data(), part.p, nblk * 8, hipMemcpyDeviceToHost, cx.stream));
    RS_TRY(hipStreamSynchronize(cx.stream));
    out = fold_partials(h);
    return true;
}

// kde::Builder{monotonic, bins, bw_adjust = x * bw_mult}.build(scores, decoys)  (kde.rs:85-133)
bool kde_build(Ctx& cx, const double* d_scores, const uint8_t* d_decoy, uint64_t n, bool monotonic, uint32_t nbins,
               double bw_mult, KdeFit& fit) {
    // the class vectors `d` and `t` (kde.rs:86-98), each in input order
    Buf<uint32_t> flags, dpos;
    Buf<double> xs, mm;
    RS_TRY(flags.alloc(n));
    RS_TRY(dpos.alloc(n));
    RS_TRY(xs.alloc(n));
    class_flags_kernel<<<grid_for(n, RB), RB, 0, cx.stream>>>(d_decoy, n, flags.p);
    if (!exclusive_count(cx, flags.p, dpos.p, n)) return false;
    uint32_t last_pos = 0, last_flag = 0;
    RS_TRY(hipMemcpyAsync(&last_pos, dpos.p + (n - 1), 4, hipMemcpyDeviceToHost, cx.stream));
    RS_TRY(hipMemcpyAsync(&last_flag, flags.p + (n - 1), 4, hipMemcpyDeviceToHost, cx.stream));
    const uint32_t nb = blocks_for(n);
    RS_TRY(mm.alloc((size_t)nb * 2));
    minmax_kernel<<<nb, RB, 0, cx.stream>>>(d_scores, n, mm.p);
    std::vector<double> hmm((size_t)nb * 2);
    RS_TRY(hipMemcpyAsync(hmm.data(), mm.p, hmm.size() * 8, hipMemcpyDeviceToHost, cx.stream));
    RS_TRY(hipStreamSynchronize(cx.stream));
    const uint64_t cnt_u[2] = {(uint64_t)last_pos + last_flag, n - ((uint64_t)last_pos + last_flag)};
    class_split_kernel<<<grid_for(n, RB), RB, 0, cx.stream>>>(d_scores, d_decoy, dpos.p, n, cnt_u[0], xs.p);
    double mn = std::numeric_limits<double>::max(), mx = std::numeric_limits<double>::lowest();
    for (uint32_t b = 0; b < nb; ++b) {
        mn = std::fmin(mn, hmm[b * 2]);
        mx = std::fmax(mx, hmm[b * 2 + 1]);
    }
    const double* cls_x[2] = {xs.p, xs.p + cnt_u[0]};
    // Kde::new (kde.rs:21-32) over mean / std of ml/mod.rs:22-30
    double bw[2], constant[2];
    for (int c = 0; c < 2; ++c) {
        const double cnt = (double)cnt_u[c];
        double sum = 0.0, ss = 0.0;
        if (!blocked_sum(cx, cls_x[c], cnt_u[c], 0, 0.0, sum)) return false;
        const double mean = sum / cnt;
        if (!blocked_sum(cx, cls_x[c], cnt_u[c], 1, mean, ss)) return false;
        const double sigma = std::sqrt(ss / cnt);
        bw[c] = (sigma * std::pow((4.0 / 3.0) / cnt, 1.0 / 5.0)) * bw_mult;
        constant[c] = std::sqrt(2.0 * M_PI) * bw[c] * cnt;
    }
    const double pi = (double)cnt_u[0] / (double)n;
    const double step = (mx - mn) / (double)(nbins - 1);
    const uint32_t nblk[2] = {(uint32_t)((cnt_u[0] + DB - 1) / DB), (uint32_t)((cnt_u[1] + DB - 1) / DB)};
    Buf<double> pdf_partial[2];
    RS_TRY(fit.bins.alloc(nbins));
    for (int c = 0; c < 2; ++c) {
        RS_TRY(pdf_partial[c].alloc((size_t)nblk[c] * nbins));
        if (nblk[c])
            kde_pdf_kernel<<<dim3(grid_for(nbins, KDE_THREADS), nblk[c]), KDE_THREADS, 0, cx.stream>>>(cls_x[c], cnt_u[c], mn, step,
                                                                                                   nbins, bw[c], pdf_partial[c].p);
    }
    kde_finish_kernel<<<1, 1024, 0, cx.stream>>>(pdf_partial[0].p, nblk[0], pdf_partial[1].p, nblk[1], nbins, constant[0],
                                                constant[1], pi, monotonic ? 1 : 0, fit.bins.p);
    RS_TRY(hipGetLastError());
    fit.dev = KdeDev{fit.bins.p, mn, step, nbins};
    return true;
}

// Gauss::solve (gauss.rs:27-165) for the 20 x 20 system: the identical elimination order and epsilon ladder, because the
// solution of the regularised system depends on both
struct Dense {
    std::vector<double> a;
    size_t rows, cols;
    Dense(size_t r, size_t c) : a(r * c, 0.0), rows(r), cols(c) {}
    double& at(size_t i, size_t j) { return a[i * cols + j]; }
    void swap_rows(size_t i, size_t j) {
        for (size_t k = 0; k < cols; ++k) std::swap(at(i, k), at(j, k));
    }
};

bool gauss_attempt(Dense left, Dense right, double eps, std::vector<double>& out) {
    const size_t m = left.rows, n = left.cols;
    for (size_t i = 0; i < n; ++i) left.at(i, i) += eps;  // fill_zero, :62-66
    size_t h = 0, k = 0;                                  // echelon, :89-124
    while (h < m && k < n) {
        size_t piv = 0;
        double best = std::numeric_limits<double>::lowest();
        for (size_t i = h; i < m; ++i)
            if (left.at(i, k) >= best) {
                piv = i;
                best = left.at(i, k);
            }
        if (left.at(piv, k) == 0.0) {
            ++k;
            continue;
        }
        if (h != piv) {
            left.swap_rows(h, piv);
            right.swap_rows(h, piv);
        }
        for (size_t i = h + 1; i < m; ++i) {
            const double factor = left.at(i, k) / left.at(h, k);
            left.at(i, k) = 0.0;
            for (size_t j = k + 1; j < n; ++j) left.at(i, j) -= left.at(h, j) * factor;
            for (size_t j = 0; j < right.cols; ++j) right.at(i, j) -= right.at(h, j) * factor;
        }
        ++h;
        ++k;
    }
    for (size_t i = m; i-- > 0;)  // reduce, :127-143
        for (size_t j = 0; j < n; ++j) {
            const double x = left.at(i, j);
            if (x == 0.0) continue;
            for (size_t c = j; c < n; ++c) left.at(i, c) /= x;
            for (size_t c = 0; c < right.cols; ++c) right.at(i, c) /= x;
            break;
        }
    for (size_t i = m; i-- > 0;)  // backfill, :146-164
        for (size_t j = 0; j < n; ++j) {
            if (left.at(i, j) == 0.0) continue;
            for (size_t r = 0; r < i; ++r) {
                const double factor = left.at(r, j) / left.at(i, j);
                for (size_t c = 0; c < n; ++c) left.at(r, c) -= left.at(i, c) * factor;
                for (size_t c = 0; c < right.cols; ++c) right.at(r, c) -= right.at(i, c) * factor;
            }
            break;
        }
    for (size_t i = 0; i < n; ++i)  // left_solved, :69-87
        for (size_t j = 0; j < n; ++j) {
            const double x = left.at(i, j);
            if (i == j) {
                if (x != 1.0 && x != 0.0) return false;
            } else if (x > 1e-8) {
                return false;
            }
        }
    out = right.a;
    return true;
}

bool gauss_solve(const Dense& left, const Dense& right, std::vector<double>& out) {  // :43-52
    for (double eps = 1e-8; eps <= 1.0; eps *= 10.0)
        if (gauss_attempt(left, right, eps, out)) return true;
    return false;
}

// descending stable sort of `n` f32 scores given as total-order keys; order_out[j] = index of the j-th best
bool sort_desc(Ctx& cx, uint32_t* d_keys_in, uint32_t* d_idx_in, uint32_t n, uint32_t* d_keys_out, uint32_t* d_idx_out) {
    size_t temp_bytes = 0;
    RS_TRY(rocprim::radix_sort_pairs_desc((void*)nullptr, temp_bytes, d_keys_in, d_keys_out, d_idx_in, d_idx_out, n, 0, 32,
                                          cx.stream));
    Buf<uint8_t> temp;
    RS_TRY(temp.alloc(temp_bytes));
    RS_TRY(rocprim::radix_sort_pairs_desc((void*)temp.p, temp_bytes, d_keys_in, d_keys_out, d_idx_in, d_idx_out, n, 0, 32,
                                          cx.stream));
    RS_TRY(hipStreamSynchronize(cx.stream));
    return true;
}

// out[j] = sum of flags[0..j]
bool prefix_count(Ctx& cx, const uint32_t* d_flags, uint32_t* d_out, uint32_t n) {
    size_t temp_bytes = 0;
    RS_TRY(rocprim::inclusive_scan((void*)nullptr, temp_bytes, d_flags, d_out, (size_t)n, rocprim::plus<uint32_t>(), cx.stream));
    Buf<uint8_t> temp;
    RS_TRY(temp.alloc(temp_bytes));
    RS_TRY(rocprim::inclusive_scan((void*)temp.p, temp_bytes, d_flags, d_out, (size_t)n, rocprim::plus<uint32_t>(), cx.stream));
    RS_TRY(hipStreamSynchronize(cx.stream));
    return true;
}

// out[j] = sum of flags[0..j)
bool exclusive_count(Ctx& cx, const uint32_t* d_flags, uint32_t* d_out, uint64_t n) {
    size_t temp_bytes = 0;
    RS_TRY(rocprim::exclusive_scan((void*)nullptr, temp_bytes, d_flags, d_out, 0u, (size_t)n, rocprim::plus<uint32_t>(), cx.stream));
    Buf<uint8_t> temp;
    RS_TRY(temp.alloc(temp_bytes));
    RS_TRY(rocprim::exclusive_scan((void*)temp.p, temp_bytes, d_flags, d_out, 0u, (size_t)n, rocprim::plus<uint32_t>(), cx.stream));
    return true;
}

// out[j] = min(q[j], q[j + 1], ..., q[n - 1]): the reverse cumulative minimum of qvalue.rs:27-35 / fdr.rs:104-112
bool suffix_min(Ctx& cx, const float* d_q, float* d_out, uint32_t n) {
    if (n == 0) return true;
    auto in = rocprim::make_reverse_iterator(d_q + n);
    auto out = rocprim::make_reverse_iterator(d_out + n);
    size_t temp_bytes = 0;
    RS_TRY(rocprim::inclusive_scan((void*)nullptr, temp_bytes, in, out, (size_t)n, OpFmin(), cx.stream));
    Buf<uint8_t> temp;
    RS_TRY(temp.alloc(temp_bytes));
    RS_TRY(rocprim::inclusive_scan((void*)temp.p, temp_bytes, in, out, (size_t)n, OpFmin(), cx.stream));
    RS_TRY(hipStreamSynchronize(cx.stream));
    return true;
}

// Competition::assign_q_value over dense keys (fdr.rs:60-120) and the write-back of fdr.rs:146-148 / :179-185
bool picked(Ctx& cx, const uint32_t* d_key, uint32_t n_keys, const uint8_t* d_decoy, const float* d_score, uint64_t n,
            float* d_q_out, uint64_t& passing) {
    passing = 0;
    if (n_keys == 0) {  // no feature takes part: every q stays 1.0
        Buf<float> dummy;
        RS_TRY(dummy.alloc(2));
        picked_gather_kernel<<<grid_for(n, RB), RB, 0, cx.stream>>>(d_key, d_decoy, n, dummy.p, d_q_out);
        RS_TRY(hipStreamSynchronize(cx.stream));
        return true;
    }
    const uint32_t n_rows = 2 * n_keys;
    Buf<uint32_t> side, row_key, row_id, sorted_key, sorted_id, counters;
    Buf<double> winner;
    Buf<uint8_t> winner_decoy, row_decoy;
    Buf<float> pep, dsum, q, qmin, side_q;
    Buf<uint32_t> target_flag, target_cum;
    Buf<unsigned long long> d_pass;
    RS_TRY(side.alloc(n_rows));
    RS_TRY(row_key.alloc(n_rows));
    RS_TRY(row_id.alloc(n_rows));
    RS_TRY(sorted_key.alloc(n_rows));
    RS_TRY(sorted_id.alloc(n_rows));
    RS_TRY(counters.alloc(2));
    RS_TRY(winner.alloc(n_keys));
    RS_TRY(winner_decoy.alloc(n_keys));
    RS_TRY(row_decoy.alloc(n_rows));
    RS_TRY(pep.alloc((size_t)n_rows + CS_TILE));
    RS_TRY(dsum.alloc((size_t)n_rows + CS_TILE));
    RS_TRY(q.alloc(n_rows));
    RS_TRY(qmin.alloc(n_rows));
    RS_TRY(target_flag.alloc(n_rows));
    RS_TRY(target_cum.alloc(n_rows));
    RS_TRY(side_q.alloc(n_rows));
    RS_TRY(d_pass.alloc(1));
    RS_TRY(hipMemsetAsync(side.p, 0, (size_t)n_rows * 4, cx.stream));
    RS_TRY(hipMemsetAsync(counters.p, 0, 8, cx.stream));
    picked_max_kernel<<<grid_for(n, RB), RB, 0, cx.stream>>>(d_key, d_decoy, d_score, n, side.p);
    picked_rows_kernel<<<grid_for(n_keys, RB), RB, 0, cx.stream>>>(side.p, n_keys, winner.p, winner_decoy.p, row_key.p, row_id.p,
                                                                    counters.p);
    uint32_t h_counters[2];
    RS_TRY(hipMemcpyAsync(h_counters, counters.p, 8, hipMemcpyDeviceToHost, cx.stream));
    RS_TRY(hipStreamSynchronize(cx.stream));
    if (h_counters[1]) {
        cx.code = SAGE_HIP_ERR_INVALID;
        cx.err = "sage_hip_rescore: competition keys must be dense (" + std::to_string(h_counters[1]) + " of " +
                 std::to_string(n_keys) + " ids are not used by any feature)";
        return false;
    }
    const uint32_t m = h_counters[0];
    KdeFit est;
    if (!kde_build(cx, winner.p, winner_decoy.p, n_keys, true, 1000, 1.0, est)) return false;
    if (!sort_desc(cx, row_key.p, row_id.p, n_rows, sorted_key.p, sorted_id.p)) return false;  // absent rows (key 0) sort last
    const uint32_t m_padded = (m + CS_TILE - 1u) / CS_TILE * CS_TILE;
    picked_pep_kernel<<<grid_for(m_padded, RB), RB, 0, cx.stream>>>(sorted_key.p, sorted_id.p, m, m_padded, est.dev, pep.p,
                                                                     row_decoy.p, target_flag.p);
    if (!prefix_count(cx, target_flag.p, target_cum.p, m)) return false;
    seq_cumsum_kernel<<<1, 64, 0, cx.stream>>>(pep.p, m_padded, dsum.p);
    picked_q_kernel<<<grid_for(m, RB), RB, 0, cx.stream>>>(dsum.p, target_cum.p, m, q.p);
    if (!suffix_min(cx, q.p, qmin.p, m)) return false;
    RS_TRY(hipMemsetAsync(d_pass.p, 0, 8, cx.stream));
    count_passing_kernel<<<grid_for(m, RB), RB, 0, cx.stream>>>(qmin.p, m, row_decoy.p, 0.01f, d_pass.p);
    // rows that do not exist keep q = 1.0 (never read: a feature's own side always exists)
    std::vector<float> ones(n_rows, 1.0f);
    RS_TRY(hipMemcpyAsync(side_q.p, ones.data(), (size_t)n_rows * 4, hipMemcpyHostToDevice, cx.stream));
    picked_scatter_kernel<<<grid_for(m, RB), RB, 0, cx.stream>>>(sorted_id.p, qmin.p, m, side_q.p);
    picked_gather_kernel<<<grid_for(n, RB), RB, 0, cx.stream>>>(d_key, d_decoy, n, side_q.p, d_q_out);
    unsigned long long h_pass = 0;
    RS_TRY(hipMemcpyAsync(&h_pass, d_pass.p, 8, hipMemcpyDeviceToHost, cx.stream));
    RS_TRY(hipGetLastError());
    RS_TRY(hipStreamSynchronize(cx.stream));
    passing = h_pass;
    return true;
}

bool rescore_impl(Ctx& cx, const SageRescoreInput& in, SageRescoreOutput& out) {
    const uint64_t n = in.n;
    const int tol_kind = in.precursor_tol.kind;
    Buf<SageFeature> feats;
    Buf<uint8_t> decoy;
    Buf<double> dmass, rows, disc, partial, folded;
    Buf<float> a_rt, d_rt, d_ims, discriminant, posterior, spectrum_q, peptide_q, protein_q, qsorted, qmin_sorted;
    Buf<uint32_t> pkey, prkey, keys, idx, keys_sorted, order;
    Buf<unsigned long long> d_pass;
    RS_TRY(feats.alloc(n));
    RS_TRY(decoy.alloc(n));
    RS_TRY(dmass.alloc(n));
    RS_TRY(rows.alloc(n * NF));
    RS_TRY(disc.alloc(n));
    RS_TRY(discriminant.alloc(n));
    RS_TRY(posterior.alloc(n));
    RS_TRY(spectrum_q.alloc(n));
    RS_TRY(peptide_q.alloc(n));
    RS_TRY(protein_q.alloc(n));
    RS_TRY(qsorted.alloc(n));
    RS_TRY(qmin_sorted.alloc(n));
    RS_TRY(pkey.alloc(n));
    RS_TRY(prkey.alloc(n));
    RS_TRY(keys.alloc(n));
    RS_TRY(idx.alloc(n));
    RS_TRY(keys_sorted.alloc(n));
    RS_TRY(order.alloc(n));
    RS_TRY(d_pass.alloc(1));
    RS_TRY(hipMemcpyAsync(feats.p, in.features, n * sizeof(SageFeature), hipMemcpyHostToDevice, cx.stream));
    RS_TRY(hipMemcpyAsync(pkey.p, in.peptide_key, n * 4, hipMemcpyHostToDevice, cx.stream));
    RS_TRY(hipMemcpyAsync(prkey.p, in.protein_key, n * 4, hipMemcpyHostToDevice, cx.stream));
    const float* opt[3] = {in.aligned_rt, in.delta_rt_model, in.delta_ims_model};
    Buf<float>* optbuf[3] = {&a_rt, &d_rt, &d_ims};
    for (int k = 0; k < 3; ++k)
        if (opt[k]) {
            RS_TRY(optbuf[k]->alloc(n));
            RS_TRY(hipMemcpyAsync(optbuf[k]->p, opt[k], n * 4, hipMemcpyHostToDevice, cx.stream));
        }
    EventPair ev;
    RS_TRY(hipEventCreate(&ev.start));
    RS_TRY(hipEventCreate(&ev.stop));
    RS_TRY(hipEventRecord(ev.start, cx.stream));
    const uint32_t g = grid_for(n, RB);
    prep_kernel<<<g, RB, 0, cx.stream>>>(feats.p, n, tol_kind, decoy.p, dmass.p);

    // ---- score_psms (linear_discriminant.rs:133-231) ----
    const double bw_adjust = tol_kind == SAGE_TOL_PPM ? 2.0 : 0.1;  // :146-150
    const float span = in.precursor_tol.hi - in.precursor_tol.lo;
    const float bin_size = tol_kind == SAGE_TOL_PPM ? std::fmax(span, 100.0f) : std::fmax(span, 1000.0f);
    const uint32_t mass_bins = (uint32_t)std::fabs(std::ceil(bin_size));
    KdeFit mass_model;
    if (!kde_build(cx, dmass.p, decoy.p, n, false, mass_bins, bw_adjust, mass_model)) return false;
    rows_kernel<<<g, RB, 0, cx.stream>>>(feats.p, n, dmass.p, mass_model.dev, a_rt.p, d_rt.p, d_ims.p, rows.p);

    // train (:57-127): class sums -> means -> scatter -> solve; both passes in the reference's row order (seq_lda_kernel)
    RS_TRY(folded.alloc(2 * NF * NF + 2 * NF));
    seq_lda_kernel<false><<<dim3(1, 2), SEQ_THREADS, 0, cx.stream>>>(rows.p, decoy.p, n, nullptr, folded.p);
    double class_sum[2][NF];
    RS_TRY(hipMemcpyAsync(class_sum, folded.p, sizeof(class_sum), hipMemcpyDeviceToHost, cx.stream));
    // class counts: the decoy flags summed by the mass-model fit would do; recount on the host from the labels instead
    RS_TRY(hipStreamSynchronize(cx.stream));
    uint64_t class_count[2] = {0, 0};
    for (uint64_t i = 0; i < n; ++i) class_count[in.features[i].label == -1 ? 0 : 1]++;
    bool fitted = class_count[0] != 0 && class_count[1] != 0;  // :83-85
    std::vector<double> coef;
    if (fitted) {
        double class_mean[2][NF];
        for (int c = 0; c < 2; ++c)
            for (int j = 0; j < NF; ++j) class_mean[c][j] = class_sum[c][j] / (double)class_count[c];
        double* d_mean = folded.p + 2 * NF * NF;
        RS_TRY(hipMemcpyAsync(d_mean, class_mean, sizeof(class_mean), hipMemcpyHostToDevice, cx.stream));
        seq_lda_kernel<true><<<dim3(1, 2), SEQ_THREADS, 0, cx.stream>>>(rows.p, decoy.p, n, d_mean, folded.p);
        std::vector<double> scatter(2 * NF * NF);
        RS_TRY(hipMemcpyAsync(scatter.data(), folded.p, scatter.size() * 8, hipMemcpyDeviceToHost, cx.stream));
        RS_TRY(hipStreamSynchronize(cx.stream));
        Dense within(NF, NF), mu(NF, 1);
        for (int c = 0; c < 2; ++c)  // :105-108
            for (int e = 0; e < NF * NF; ++e) within.a[e] += scatter[c * NF * NF + e] / (double)class_count[c];
        for (int j = 0; j < NF; ++j) mu.a[j] = class_mean[1][j] - class_mean[0][j];
        fitted = gauss_solve(within, mu, coef);
        if (fitted)
            for (double c : coef)
                if (!std::isfinite(c)) fitted = false;  // :198-210
    }
    out.lda_fitted = fitted ? 1 : 0;
    std::memset(out.coef, 0, sizeof(out.coef));
    KdeFit kde;
    if (fitted) {
        Coef cf;
        for (int j = 0; j < NF; ++j) out.coef[j] = cf.w[j] = coef[j];
        project_kernel<<<g, RB, 0, cx.stream>>>(rows.p, n, cf, disc.p);
        if (!kde_build(cx, disc.p, decoy.p, n, true, 1000, 1.0, kde)) return false;
        pep_kernel<<<g, RB, 0, cx.stream>>>(disc.p, n, kde.dev, discriminant.p, posterior.p);
    } else {
        heuristic_kernel<<<g, RB, 0, cx.stream>>>(feats.p, n, discriminant.p, posterior.p);
    }

    // ---- runner.rs:290-291: sort by discriminant, spectrum_q_value ----
    sort_keys_kernel<<<g, RB, 0, cx.stream>>>(discriminant.p, (uint32_t)n, keys.p, idx.p);
    if (!sort_desc(cx, keys.p, idx.p, (uint32_t)n, keys_sorted.p, order.p)) return false;
    decoy_flags_kernel<<<g, RB, 0, cx.stream>>>(order.p, decoy.p, (uint32_t)n, keys.p);  // (keys / idx are free again)
    if (!prefix_count(cx, keys.p, idx.p, (uint32_t)n)) return false;
    q_from_counts_kernel<<<g, RB, 0, cx.stream>>>(idx.p, (uint32_t)n, qsorted.p);
    if (!suffix_min(cx, qsorted.p, qmin_sorted.p, (uint32_t)n)) return false;
    RS_TRY(hipMemsetAsync(d_pass.p, 0, 8, cx.stream));
    count_passing_kernel<<<g, RB, 0, cx.stream>>>(qmin_sorted.p, (uint32_t)n, nullptr, 0.01f, d_pass.p);
    scatter_by_order_kernel<<<g, RB, 0, cx.stream>>>(order.p, qmin_sorted.p, (uint32_t)n, spectrum_q.p);
    unsigned long long h_pass = 0;
    RS_TRY(hipMemcpyAsync(&h_pass, d_pass.p, 8, hipMemcpyDeviceToHost, cx.stream));
    RS_TRY(hipGetLastError());
    RS_TRY(hipStreamSynchronize(cx.stream));
    out.passing_spectrum = h_pass;

    // ---- fdr.rs:123-187 ----
    if (!picked(cx, pkey.p, in.n_peptide_keys, decoy.p, discriminant.p, n, peptide_q.p, out.passing_peptide)) return false;
    if (!picked(cx, prkey.p, in.n_protein_keys, decoy.p, discriminant.p, n, protein_q.p, out.passing_protein)) return false;
    RS_TRY(hipEventRecord(ev.stop, cx.stream));

    RS_TRY(hipMemcpyAsync(out.discriminant_score, discriminant.p, n * 4, hipMemcpyDeviceToHost, cx.stream));
    RS_TRY(hipMemcpyAsync(out.posterior_error, posterior.p, n * 4, hipMemcpyDeviceToHost, cx.stream));
    RS_TRY(hipMemcpyAsync(out.spectrum_q, spectrum_q.p, n * 4, hipMemcpyDeviceToHost, cx.stream));
    RS_TRY(hipMemcpyAsync(out.peptide_q, peptide_q.p, n * 4, hipMemcpyDeviceToHost, cx.stream));
    RS_TRY(hipMemcpyAsync(out.protein_q, protein_q.p, n * 4, hipMemcpyDeviceToHost, cx.stream));
    if (out.order) RS_TRY(hipMemcpyAsync(out.order, order.p, n * 4, hipMemcpyDeviceToHost, cx.stream));
    RS_TRY(hipStreamSynchronize(cx.stream));
    float ms = 0.0f;
    RS_TRY(hipEventElapsedTime(&ms, ev.start, ev.stop));
    out.device_ms = ms;
    return true;
}


// ======================================================================================================================
// The predict_rt block (sage-cli runner.rs:513-530): poisson-sorted q-values -> global_alignment -> retention / mobility
// linear models.  Same division of labour as above: everything per-PSM or per-(PSM x feature) is a kernel, the D x D solve
// and a handful of scalars are host arithmetic.
// ======================================================================================================================

constexpr int RT_D = 22 * 3 + 3;   // retention_model.rs:33
constexpr int IM_D = 22 * 4 + 12;  // mobility_model.rs:78
__constant__ uint8_t kAaMap[26] = {0, 0, 1, 2, 3, 4, 5, 6, 7, 0, 8, 9, 10, 11, 21, 12, 13, 14, 15, 16, 20, 17, 18, 0, 19, 0};
// ^ retention_model.rs:65-68 over mass.rs:59-62 VALID_AA = ACDEFGHIKLMNPQRSTVWYUO (letters outside it map to 0)

// RetentionModel::embed (retention_model.rs:44-62) into e[RT_D] (any addressable memory)
struct RtEmbed {
    static constexpr int D = RT_D;
    __device__ static void embed(const uint8_t* __restrict__ seq, uint32_t len, float mono, uint8_t, double* e) {
        for (int j = 0; j < D; ++j) e[j] = 0.0;
        const uint32_t cterm = len >= 3 ? len - 3 : 0;
        for (uint32_t a = 0; a < len; ++a) {
            const uint32_t idx = kAaMap[(uint8_t)(seq[a] - 'A') < 26 ? seq[a] - 'A' : 0];
            e[idx] += 1.0;
            if (a == 0 || a == 1) e[22 + idx] += 1.0;
            else if (a == cterm || a == cterm + 1) e[44 + idx] += 1.0;
        }
        e[D - 3] = (double)len;
        e[D - 2] = log1p((double)mono);
        e[D - 1] = 1.0;
    }
    __device__ static double target(const SageFeature& f, float aligned_rt) { return (double)aligned_rt; }
};

// MobilityModel::embed (mobility_model.rs:103-158).  The residue-class tables of the reference hold LETTER offsets
// (b'L' - b'A', ...) but are compared with the residue's VALID_AA index (:121-139); restated as written:
// bulky {11,21,8,5,22,24}, uncharged polar {18,19,13,16}, positive {17,10,7}, negative {3,4}, tiny {6,0,18}, branched {11,8,21}.
struct ImEmbed {
    static constexpr int D = IM_D;
    __device__ static void embed(const uint8_t* __restrict__ seq, uint32_t len, float mono, uint8_t charge, double* e) {
        for (int j = 0; j < D; ++j) e[j] = 0.0;
        const uint32_t cterm = len >= 3 ? len - 3 : 0;
        for (uint32_t a = 0; a < len; ++a) {
            const uint32_t x = kAaMap[(uint8_t)(seq[a] - 'A') < 26 ? seq[a] - 'A' : 0];
            e[x] += 1.0;
            if (a == 0 || a == 1) e[44 + x] += 1.0;
            else if (a > cterm) e[66 + x] += 1.0;
            if (x == 11 || x == 21 || x == 8 || x == 5) e[D - 9] += 1.0;    // NUM_BULKY
            if (x == 18 || x == 19 || x == 13 || x == 16) e[D - 10] += 1.0;  // NUM_UC_POLAR
            if (x == 17 || x == 10 || x == 7) e[D - 8] += 1.0;               // NUM_POSITIVE
            if (x == 3 || x == 4) e[D - 7] += 1.0;                           // NUM_NEGATIVE
            if (x == 6 || x == 0 || x == 18) e[D - 11] += 1.0;               // NUM_TINY
            if (x == 11 || x == 8 || x == 21) e[D - 12] += 1.0;              // NUM_BRANCHED
        }
        for (int i = 0; i < 22; ++i) e[22 + i] = e[i] / (double)len;
        const double z = (double)charge;
        e[D - 5] = z;
        e[D - 6] = 1.0 / z;
        e[D - 3] = (double)len;
        e[D - 2] = (double)mono / 1000.0;
        e[D - 4] = ((double)mono / z) / 1000.0;
        e[D - 1] = 1.0;
    }
    __device__ static double target(const SageFeature& f, float) { return (double)f.ims; }
};

__device__ inline uint64_t total_order_key64(double x) {  // ascending u64 order == f64::total_cmp
    const uint64_t b = (uint64_t)__double_as_longlong(x);
    return b ^ ((b >> 63) ? 0xFFFFFFFFFFFFFFFFull : 0x8000000000000000ull);
}

__global__ __launch_bounds__(RB) void poisson_keys_kernel(const SageFeature* __restrict__ f, uint32_t n, uint64_t* __restrict__ keys,
                                                          uint32_t* __restrict__ idx, uint8_t* __restrict__ decoy) {
    const uint32_t i = blockIdx.x * RB + threadIdx.x;
    if (i >= n) return;
    keys[i] = total_order_key64(f[i].poisson);
    idx[i] = i;
    decoy[i] = f[i].label == -1;
}

// the training filter of both models and of the alignment: label == 1 && spectrum_q <= 0.01; max_rt_by_file
// (retention_alignment.rs:26-41: fetch_max of `rt.ceil() as u32`, a saturating cast)
__global__ __launch_bounds__(RB) void train_flags_kernel(const SageFeature* __restrict__ f, const float* __restrict__ q, uint32_t n,
                                                         uint32_t n_files, uint8_t* __restrict__ train,
                                                         uint32_t* __restrict__ max_rt, uint32_t* __restrict__ bad_file) {
    const uint32_t i = blockIdx.x * RB + threadIdx.x;
    if (i >= n) return;
    train[i] = f[i].label == 1 && q[i] <= 0.01f;
    if (f[i].file_id >= n_files) {
        atomicAdd(bad_file, 1u);
        return;
    }
    const float c = ceilf(f[i].rt);
    const uint32_t v = (!(c == c) || c <= 0.0f) ? 0u : (c >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)c);
    atomicMax(&max_rt[f[i].file_id], v);
}

// sort keys of the training PSMs for the (peptide, file) grouping of mean_rt_by_file (:45-60); others sort last
__global__ __launch_bounds__(RB) void group_keys_kernel(const SageFeature* __restrict__ f, const uint8_t* __restrict__ train, uint32_t n,
                                                        uint64_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    const uint32_t i = blockIdx.x * RB + threadIdx.x;
    if (i >= n) return;
    keys[i] = train[i] ? ((uint64_t)f[i].peptide_idx << 32) | f[i].file_id : 0xFFFFFFFFFFFFFFFFull;
    idx[i] = i;
}

// row_start[j] = 1 where a new peptide begins among the first m (training) sorted entries
__global__ __launch_bounds__(RB) void row_start_kernel(const uint64_t* __restrict__ keys, uint32_t m, uint32_t* __restrict__ row_start) {
    const uint32_t j = blockIdx.x * RB + threadIdx.x;
    if (j < m) row_start[j] = (j == 0 || (keys[j] >> 32) != (keys[j - 1] >> 32)) ? 1u : 0u;
}

// the head of every (peptide, file) run writes the run's minimum rt, divided by the file's maximum (rt_matrix, :62-90),
// into mat[row][file]; mat is pre-filled with NaN
__global__ __launch_bounds__(RB) void run_min_kernel(const SageFeature* __restrict__ f, const uint64_t* __restrict__ keys,
                                                     const uint32_t* __restrict__ idx, const uint32_t* __restrict__ row_cum, uint32_t m,
                                                     uint32_t n_files, const double* __restrict__ max_rt, double* __restrict__ mat) {
    const uint32_t j = blockIdx.x * RB + threadIdx.x;
    if (j >= m || (j > 0 && keys[j] == keys[j - 1])) return;
    const uint64_t key = keys[j];
    double mn = (double)f[idx[j]].rt;
    for (uint32_t k = j + 1; k < m && keys[k] == key; ++k) mn = fmin(mn, (double)f[idx[k]].rt);  // f64::min
    const uint32_t file = (uint32_t)key;
    mat[(uint64_t)(row_cum[j] - 1) * n_files + file] = mn / max_rt[file];
}

// per row: the mean over the files that saw the peptide; rows whose mean is not a normal number are dropped (:79) by
// turning them into all-NaN rows; mean_rts (:104-115)
__global__ __launch_bounds__(RB) void row_mean_kernel(double* __restrict__ mat, uint32_t n_rows, uint32_t n_files,
                                                      double* __restrict__ mean_rts) {
    const uint32_t r = blockIdx.x * RB + threadIdx.x;
    if (r >= n_rows) return;
    double sum = 0.0, len = 0.0;
    for (uint32_t k = 0; k < n_files; ++k) {
        const double v = mat[(uint64_t)r * n_files + k];
        if (v == v) {
            sum += v;
            len += 1.0;
        }
    }
    const double mean = sum / len, a = fabs(mean);
    const bool normal = mean == mean && a >= 2.2250738585072014e-308 && a <= 1.7976931348623157e308;
    if (!normal)
        for (uint32_t k = 0; k < n_files; ++k) mat[(uint64_t)r * n_files + k] = __longlong_as_double(0x7FF8000000000000ll);
    // (a finite entry count recomputed as in :108-113; infinite entries can only come from max_rt == 0)
    double s2 = 0.0;
    uint32_t l2 = 0;
    for (uint32_t k = 0; k < n_files; ++k) {
        const double v = mat[(uint64_t)r * n_files + k];
        if (isfinite(v)) {
            s2 += v;
            ++l2;
        }
    }
    mean_rts[r] = s2 / (double)l2;
}

// per file (blockIdx.y): partial {len, dot, sum_x, sum_y} (pass 0) or {sum (x - x_mean)^2} (pass 1) over the finite entries
// of the file's column (:121-141)
__global__ __launch_bounds__(RB) void align_sums_kernel(const double* __restrict__ mat, const double* __restrict__ mean_rts,
                                                        uint32_t n_rows, uint32_t n_files, int pass, const double* __restrict__ x_mean,
                                                        double* __restrict__ partial) {
    __shared__ double lds[RB / 64];
    const uint32_t file = blockIdx.y;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (uint32_t r = blockIdx.x * RB + threadIdx.x; r < n_rows; r += gridDim.x * RB) {
        const double x = mat[(uint64_t)r * n_files + file];
        if (!isfinite(x)) continue;
        if (pass == 0) {
            const double y = mean_rts[r];
            a0 += 1.0;
            a1 += x * y;
            a2 += x;
            a3 += y;
        } else {
            const double d = x - x_mean[file];
            a0 += d * d;
        }
    }
    const double r0 = block_reduce(a0, OpSum(), lds), r1 = block_reduce(a1, OpSum(), lds), r2 = block_reduce(a2, OpSum(), lds),
                 r3 = block_reduce(a3, OpSum(), lds);
    if (threadIdx.x == 0) {
        double* o = partial + ((uint64_t)blockIdx.x * n_files + file) * 4;
        o[0] = r0;
        o[1] = r1;
        o[2] = r2;
        o[3] = r3;
    }
}

// feature.aligned_rt = (feature.rt / a.max_rt) * a.slope + a.intercept, in f32 (:165-172)
__global__ __launch_bounds__(RB) void aligned_rt_kernel(const SageFeature* __restrict__ f, uint32_t n, const SageAlignment* __restrict__ al,
                                                        float* __restrict__ aligned) {
    const uint32_t i = blockIdx.x * RB + threadIdx.x;
    if (i >= n) return;
    const SageAlignment a = al[f[i].file_id];
    aligned[i] = (f[i].rt / a.max_rt) * a.slope + a.intercept;
}

// LinearRegression::fit pass 1 (regression.rs:68-84): partial[b] = {X^T X (D*D), X^T y (D), sum y, sum y^2, count} over the
// block's training rows.  Rows are embedded 16 at a time into LDS (one thread per row), then every thread adds its share
// of the D*D + D + 3 accumulators.
constexpr int XTX_THREADS = 1024, XTX_STAGE = 16;
template <class E>
__global__ __launch_bounds__(XTX_THREADS) void xtx_kernel(const SageFeature* __restrict__ f, const uint8_t* __restrict__ train,
                                                          const uint64_t* __restrict__ seq_off, const uint8_t* __restrict__ seq,
                                                          const float* __restrict__ mono, const float* __restrict__ aligned,
                                                          uint32_t n, double* __restrict__ partial) {
    constexpr int D = E::D, W = D * D + D + 3, PER = (W + XTX_THREADS - 1) / XTX_THREADS;
    __shared__ double stage[XTX_STAGE][D];
    __shared__ double ys[XTX_STAGE];
    __shared__ uint8_t use[XTX_STAGE];
    const uint32_t per = (n + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    double acc[PER];
#pragma unroll
    for (int s = 0; s < PER; ++s) acc[s] = 0.0;
    for (uint32_t base = lo; base < hi; base += XTX_STAGE) {
        __syncthreads();
        if (threadIdx.x < XTX_STAGE) {
            const uint32_t i = base + threadIdx.x;
            const bool u = i < hi && train[i];
            use[threadIdx.x] = u;
            if (u) {
                E::embed(seq + seq_off[i], (uint32_t)(seq_off[i + 1] - seq_off[i]), mono[i], f[i].charge, stage[threadIdx.x]);
                ys[threadIdx.x] = E::target(f[i], aligned ? aligned[i] : 0.0f);
            }
        }
        __syncthreads();
        for (int r = 0; r < XTX_STAGE; ++r) {
            if (!use[r]) continue;
            const double y = ys[r];
#pragma unroll
            for (int s = 0; s < PER; ++s) {
                const int e = threadIdx.x + s * XTX_THREADS;
                if (e < D * D) acc[s] += stage[r][e / D] * stage[r][e % D];
                else if (e < D * D + D) acc[s] += stage[r][e - D * D] * y;
                else if (e == D * D + D) acc[s] += y;
                else if (e == D * D + D + 1) acc[s] += y * y;
                else if (e == D * D + D + 2) acc[s] += 1.0;
            }
        }
    }
#pragma unroll
    for (int s = 0; s < PER; ++s) {
        const int e = threadIdx.x + s * XTX_THREADS;
        if (e < W) partial[(uint64_t)blockIdx.x * W + e] = acc[s];
    }
}

template <int D>
struct Beta {
    double w[D];
};

// predictions sum_j x_j * beta_j (left to right from 0.0: regression.rs:108, retention_model.rs:84-88); one thread per PSM,
// its embedding in LDS.  mode 0: partial[b] = sum over the block's TRAINING rows of (pred - y)^2 (the SSE pass, :104-113).
// mode 1: write the clamped prediction and |observed - prediction| (retention_model.rs:17-24 / mobility_model.rs:23-30).
constexpr int PRED_THREADS = 64;
template <class E>
__global__ __launch_bounds__(PRED_THREADS) void predict_kernel(const SageFeature* __restrict__ f, const uint8_t* __restrict__ train,
                                                               const uint64_t* __restrict__ seq_off, const uint8_t* __restrict__ seq,
                                                               const float* __restrict__ mono, const float* __restrict__ aligned,
                                                               uint32_t n, Beta<E::D> beta, int mode, double hi_clamp,
                                                               double* __restrict__ partial, float* __restrict__ predicted,
                                                               float* __restrict__ delta) {
    constexpr int D = E::D;
    __shared__ double rows[PRED_THREADS][D + 1];  // (+1: odd stride, conflict-free row-per-thread access)
    __shared__ double lds[1];
    const uint32_t i = blockIdx.x * PRED_THREADS + threadIdx.x;
    double sq = 0.0;
    if (i < n && (mode == 1 || train[i])) {
        double* e = rows[threadIdx.x];
        E::embed(seq + seq_off[i], (uint32_t)(seq_off[i + 1] - seq_off[i]), mono[i], f[i].charge, e);
        double pred = 0.0;
        for (int j = 0; j < D; ++j) pred = pred + e[j] * beta.w[j];
        const double y = E::target(f[i], aligned ? aligned[i] : 0.0f);
        if (mode == 0) {
            sq = (pred - y) * (pred - y);
        } else {
            const float bounded = (float)(pred < 0.0 ? 0.0 : (pred > hi_clamp ? hi_clamp : pred));  // f64::clamp: NaN stays NaN
            predicted[i] = bounded;
            delta[i] = fabsf((float)y - bounded);  // (y is exactly the f32 the reference subtracts from)
        }
    }
    if (mode == 0) {
        const double r = block_reduce(sq, OpSum(), lds);
        if (threadIdx.x == 0) partial[blockIdx.x] = r;
    }
}

__global__ __launch_bounds__(RB) void fill_kernel(float* __restrict__ p, uint32_t n, float v) {
    const uint32_t i = blockIdx.x * RB + threadIdx.x;
    if (i < n) p[i] = v;
}

// LinearRegression::fit + predict for one model; false only on a HIP error (an unfitted model is `fitted = false`)
template <class E>
bool fit_and_predict(Ctx& cx, const SageFeature* d_f, const uint8_t* d_train, const uint64_t* d_seq_off, const uint8_t* d_seq,
                     const float* d_mono, const float* d_aligned, uint32_t n, double hi_clamp, float* d_pred, float* d_delta,
                     bool& fitted, double& r2) {
    constexpr int D = E::D, W = D * D + D + 3;
    fitted = false;
    r2 = 0.0;
    const uint32_t nb = (uint32_t)std::min<uint64_t>(256, std::max<uint64_t>(1, (n + 255) / 256));
    Buf<double> partial, folded;
    RS_TRY(partial.alloc((size_t)nb * W));
    RS_TRY(folded.alloc(W));
    xtx_kernel<E><<<nb, XTX_THREADS, 0, cx.stream>>>(d_f, d_train, d_seq_off, d_seq, d_mono, d_aligned, n, partial.p);
    fold_partials_kernel<<<grid_for(W, RB), RB, 0, cx.stream>>>(partial.p, nb, W, folded.p);
    std::vector<double> h(W);
    RS_TRY(hipMemcpyAsync(h.data(), folded.p, (size_t)W * 8, hipMemcpyDeviceToHost, cx.stream));
    RS_TRY(hipGetLastError());
    RS_TRY(hipStreamSynchronize(cx.stream));
    const double cnt = h[D * D + D + 2];
    if (cnt == 0.0) return true;  // regression.rs:86-88
    const double sum_y = h[D * D + D], sum_y2 = h[D * D + D + 1];
    const double y_mean = sum_y / cnt, y_var = sum_y2 - cnt * y_mean * y_mean;
    Dense cov(D, D), b(D, 1);
    std::copy(h.begin(), h.begin() + D * D, cov.a.begin());
    std::copy(h.begin() + D * D, h.begin() + D * D + D, b.a.begin());
    std::vector<double> beta;
    if (!gauss_solve(cov, b, beta)) return true;  // :96
    Beta<D> bw;
    for (int j = 0; j < D; ++j) bw.w[j] = beta[j];
    const uint32_t pb = grid_for(n, PRED_THREADS);
    Buf<double> sse_partial;
    RS_TRY(sse_partial.alloc(pb));
    predict_kernel<E><<<pb, PRED_THREADS, 0, cx.stream>>>(d_f, d_train, d_seq_off, d_seq, d_mono, d_aligned, n, bw, 0, hi_clamp,
                                                           sse_partial.p, nullptr, nullptr);
    predict_kernel<E><<<pb, PRED_THREADS, 0, cx.stream>>>(d_f, d_train, d_seq_off, d_seq, d_mono, d_aligned, n, bw, 1, hi_clamp,
                                                           nullptr, d_pred, d_delta);
    std::vector<double> hs(pb);
    RS_TRY(hipMemcpyAsync(hs.data(), sse_partial.p, (size_t)pb * 8, hipMemcpyDeviceToHost, cx.stream));
    RS_TRY(hipGetLastError());
    RS_TRY(hipStreamSynchronize(cx.stream));
    double sse = 0.0;
    for (double v : hs) sse += v;
    r2 = 1.0 - sse / y_var;
    fitted = true;
    return true;
}

bool predict_rt_impl(Ctx& cx, const SageRtInput& in, SageRtOutput& out) {
    const uint32_t n = (uint32_t)in.n, nf = in.n_files;
    const uint64_t n_res = in.seq_off[n];
    Buf<SageFeature> feats;
    Buf<uint64_t> seq_off, keys, keys_sorted;
    Buf<uint8_t> seq, decoy, train;
    Buf<float> mono, q_sorted, qmin_sorted, spectrum_q, aligned, pred_rt, d_rt, pred_ims, d_ims;
    Buf<uint32_t> idx, order, flags, cum, max_rt_u, counters;
    Buf<unsigned long long> dummy_pass;
    RS_TRY(feats.alloc(n));
    RS_TRY(seq_off.alloc((size_t)n + 1));
    RS_TRY(seq.alloc(n_res));
    RS_TRY(mono.alloc(n));
    RS_TRY(keys.alloc(n));
    RS_TRY(keys_sorted.alloc(n));
    RS_TRY(idx.alloc(n));
    RS_TRY(order.alloc(n));
    RS_TRY(flags.alloc(n));
    RS_TRY(cum.alloc(n));
    RS_TRY(decoy.alloc(n));
    RS_TRY(train.alloc(n));
    RS_TRY(q_sorted.alloc(n));
    RS_TRY(qmin_sorted.alloc(n));
    RS_TRY(spectrum_q.alloc(n));
    RS_TRY(aligned.alloc(n));
    RS_TRY(pred_rt.alloc(n));
    RS_TRY(d_rt.alloc(n));
    RS_TRY(pred_ims.alloc(n));
    RS_TRY(d_ims.alloc(n));
    RS_TRY(max_rt_u.alloc(nf));
    RS_TRY(counters.alloc(1));
    RS_TRY(hipMemcpyAsync(feats.p, in.features, (size_t)n * sizeof(SageFeature), hipMemcpyHostToDevice, cx.stream));
    RS_TRY(hipMemcpyAsync(seq_off.p, in.seq_off, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, cx.stream));
    RS_TRY(hipMemcpyAsync(seq.p, in.seq, n_res, hipMemcpyHostToDevice, cx.stream));
    RS_TRY(hipMemcpyAsync(mono.p, in.monoisotopic, (size_t)n * 4, hipMemcpyHostToDevice, cx.stream));
    RS_TRY(hipMemsetAsync(max_rt_u.p, 0, (size_t)nf * 4, cx.stream));
    RS_TRY(hipMemsetAsync(counters.p, 0, 4, cx.stream));
    EventPair ev;
    RS_TRY(hipEventCreate(&ev.start));
    RS_TRY(hipEventCreate(&ev.stop));
    RS_TRY(hipEventRecord(ev.start, cx.stream));
    const uint32_t g = grid_for(n, RB);

    // ---- runner.rs:517-520: sort by poisson (f64 total order, ascending), spectrum_q_value ----
    poisson_keys_kernel<<<g, RB, 0, cx.stream>>>(feats.p, n, keys.p, idx.p, decoy.p);
    {
        size_t temp_bytes = 0;
        RS_TRY(rocprim::radix_sort_pairs((void*)nullptr, temp_bytes, keys.p, keys_sorted.p, idx.p, order.p, n, 0, 64, cx.stream));
        Buf<uint8_t> temp;
        RS_TRY(temp.alloc(temp_bytes));
        RS_TRY(rocprim::radix_sort_pairs((void*)temp.p, temp_bytes, keys.p, keys_sorted.p, idx.p, order.p, n, 0, 64, cx.stream));
        RS_TRY(hipStreamSynchronize(cx.stream));
    }
    decoy_flags_kernel<<<g, RB, 0, cx.stream>>>(order.p, decoy.p, n, flags.p);
    if (!prefix_count(cx, flags.p, cum.p, n)) return false;
    q_from_counts_kernel<<<g, RB, 0, cx.stream>>>(cum.p, n, q_sorted.p);
    if (!suffix_min(cx, q_sorted.p, qmin_sorted.p, n)) return false;
    scatter_by_order_kernel<<<g, RB, 0, cx.stream>>>(order.p, qmin_sorted.p, n, spectrum_q.p);
    train_flags_kernel<<<g, RB, 0, cx.stream>>>(feats.p, spectrum_q.p, n, nf, train.p, max_rt_u.p, counters.p);

    // ---- global_alignment (retention_alignment.rs:100-173) ----
    std::vector<uint32_t> h_max(nf);
    uint32_t h_bad = 0;
    RS_TRY(hipMemcpyAsync(h_max.data(), max_rt_u.p, (size_t)nf * 4, hipMemcpyDeviceToHost, cx.stream));
    RS_TRY(hipMemcpyAsync(&h_bad, counters.p, 4, hipMemcpyDeviceToHost, cx.stream));
    RS_TRY(hipGetLastError());
    RS_TRY(hipStreamSynchronize(cx.stream));
    if (h_bad) {
        cx.code = SAGE_HIP_ERR_INVALID;
        cx.err = "sage_hip_predict_rt: a feature's file_id is >= n_files";
        return false;
    }
    std::vector<double> h_max_d(nf);
    for (uint32_t k = 0; k < nf; ++k) h_max_d[k] = (double)h_max[k];
    Buf<double> d_max;
    RS_TRY(d_max.alloc(nf));
    RS_TRY(hipMemcpyAsync(d_max.p, h_max_d.data(), (size_t)nf * 8, hipMemcpyHostToDevice, cx.stream));
    // group the training PSMs by (peptide, file): 64-bit radix sort, run heads take the minimum
    group_keys_kernel<<<g, RB, 0, cx.stream>>>(feats.p, train.p, n, keys.p, idx.p);
    {
        size_t temp_bytes = 0;
        RS_TRY(rocprim::radix_sort_pairs((void*)nullptr, temp_bytes, keys.p, keys_sorted.p, idx.p, order.p, n, 0, 64, cx.stream));
        Buf<uint8_t> temp;
        RS_TRY(temp.alloc(temp_bytes));
        RS_TRY(rocprim::radix_sort_pairs((void*)temp.p, temp_bytes, keys.p, keys_sorted.p, idx.p, order.p, n, 0, 64, cx.stream));
        RS_TRY(hipStreamSynchronize(cx.stream));
    }
    // number of training PSMs = position of the first sentinel key; count it with the flags of the q pass
    std::vector<uint8_t> h_train(n);
    RS_TRY(hipMemcpyAsync(h_train.data(), train.p, n, hipMemcpyDeviceToHost, cx.stream));
    RS_TRY(hipStreamSynchronize(cx.stream));
    uint32_t m = 0;
    for (uint32_t i = 0; i < n; ++i) m += h_train[i];
    uint32_t n_rows = 0;
    Buf<double> mat, mean_rts;
    if (m) {
        row_start_kernel<<<grid_for(m, RB), RB, 0, cx.stream>>>(keys_sorted.p, m, flags.p);
        if (!prefix_count(cx, flags.p, cum.p, m)) return false;
        RS_TRY(hipMemcpyAsync(&n_rows, cum.p + (m - 1), 4, hipMemcpyDeviceToHost, cx.stream));
        RS_TRY(hipStreamSynchronize(cx.stream));
    }
    RS_TRY(mat.alloc((size_t)std::max<uint32_t>(n_rows, 1) * nf));
    RS_TRY(mean_rts.alloc(std::max<uint32_t>(n_rows, 1)));
    std::vector<SageAlignment> al(nf);
    {
        std::vector<double> len(nf, 0.0), dot(nf, 0.0), sum_x(nf, 0.0), sum_y(nf, 0.0), sx2(nf, 1e-8), x_mean(nf), y_mean(nf);
        if (n_rows) {
            RS_TRY(hipMemsetAsync(mat.p, 0xFF, (size_t)n_rows * nf * 8, cx.stream));  // all-ones bit pattern: a NaN
            run_min_kernel<<<grid_for(m, RB), RB, 0, cx.stream>>>(feats.p, keys_sorted.p, order.p, cum.p, m, nf, d_max.p, mat.p);
            row_mean_kernel<<<grid_for(n_rows, RB), RB, 0, cx.stream>>>(mat.p, n_rows, nf, mean_rts.p);
            const uint32_t nb = (uint32_t)std::min<uint32_t>(64, grid_for(n_rows, RB));
            Buf<double> partial, d_xmean;
            RS_TRY(partial.alloc((size_t)nb * nf * 4));
            RS_TRY(d_xmean.alloc(nf));
            std::vector<double> hp((size_t)nb * nf * 4);
            align_sums_kernel<<<dim3(nb, nf), RB, 0, cx.stream>>>(mat.p, mean_rts.p, n_rows, nf, 0, nullptr, partial.p);
            RS_TRY(hipMemcpyAsync(hp.data(), partial.p, hp.size() * 8, hipMemcpyDeviceToHost, cx.stream));
            RS_TRY(hipGetLastError());
            RS_TRY(hipStreamSynchronize(cx.stream));
            for (uint32_t b = 0; b < nb; ++b)
                for (uint32_t k = 0; k < nf; ++k) {
                    const double* o = &hp[((size_t)b * nf + k) * 4];
                    len[k] += o[0];
                    dot[k] += o[1];
                    sum_x[k] += o[2];
                    sum_y[k] += o[3];
                }
            for (uint32_t k = 0; k < nf; ++k) x_mean[k] = sum_x[k] / len[k];
            RS_TRY(hipMemcpyAsync(d_xmean.p, x_mean.data(), (size_t)nf * 8, hipMemcpyHostToDevice, cx.stream));
            align_sums_kernel<<<dim3(nb, nf), RB, 0, cx.stream>>>(mat.p, mean_rts.p, n_rows, nf, 1, d_xmean.p, partial.p);
            RS_TRY(hipMemcpyAsync(hp.data(), partial.p, hp.size() * 8, hipMemcpyDeviceToHost, cx.stream));
            RS_TRY(hipGetLastError());
            RS_TRY(hipStreamSynchronize(cx.stream));
            for (uint32_t b = 0; b < nb; ++b)
                for (uint32_t k = 0; k < nf; ++k) sx2[k] += hp[((size_t)b * nf + k) * 4];
        }
        for (uint32_t k = 0; k < nf; ++k) {  // :129-157 (0 / 0 when a file has no training peptide: slope 1, intercept 0)
            const double xm = sum_x[k] / len[k], ym = sum_y[k] / len[k];
            const double ssxy = dot[k] - len[k] * xm * ym;
            double slope = ssxy / sx2[k], intercept = ym - slope * xm;
            if (!std::isfinite(slope)) slope = 1.0;
            if (!std::isfinite(intercept)) intercept = 0.0;
            al[k] = SageAlignment{k, (float)h_max_d[k], (float)slope, (float)intercept};
        }
    }
    Buf<SageAlignment> d_al;
    RS_TRY(d_al.alloc(nf));
    RS_TRY(hipMemcpyAsync(d_al.p, al.data(), (size_t)nf * sizeof(SageAlignment), hipMemcpyHostToDevice, cx.stream));
    aligned_rt_kernel<<<g, RB, 0, cx.stream>>>(feats.p, n, d_al.p, aligned.p);

    // ---- retention_model::predict, mobility_model::predict; Feature defaults where a model is not fitted ----
    fill_kernel<<<g, RB, 0, cx.stream>>>(pred_rt.p, n, 0.0f);
    fill_kernel<<<g, RB, 0, cx.stream>>>(d_rt.p, n, 0.999f);
    fill_kernel<<<g, RB, 0, cx.stream>>>(pred_ims.p, n, 0.0f);
    fill_kernel<<<g, RB, 0, cx.stream>>>(d_ims.p, n, 0.999f);
    bool rt_ok = false, ims_ok = false;
    if (!fit_and_predict<RtEmbed>(cx, feats.p, train.p, seq_off.p, seq.p, mono.p, aligned.p, n, 1.0, pred_rt.p, d_rt.p, rt_ok,
                                  out.rt_r2))
        return false;
    if (!fit_and_predict<ImEmbed>(cx, feats.p, train.p, seq_off.p, seq.p, mono.p, nullptr, n, 2.0, pred_ims.p, d_ims.p, ims_ok,
                                  out.ims_r2))
        return false;
    out.rt_fitted = rt_ok;
    out.ims_fitted = ims_ok;
    RS_TRY(hipEventRecord(ev.stop, cx.stream));
    RS_TRY(hipMemcpyAsync(out.spectrum_q, spectrum_q.p, (size_t)n * 4, hipMemcpyDeviceToHost, cx.stream));
    RS_TRY(hipMemcpyAsync(out.aligned_rt, aligned.p, (size_t)n * 4, hipMemcpyDeviceToHost, cx.stream));
    RS_TRY(hipMemcpyAsync(out.predicted_rt, pred_rt.p, (size_t)n * 4, hipMemcpyDeviceToHost, cx.stream));
    RS_TRY(hipMemcpyAsync(out.delta_rt_model, d_rt.p, (size_t)n * 4, hipMemcpyDeviceToHost, cx.stream));
    RS_TRY(hipMemcpyAsync(out.predicted_ims, pred_ims.p, (size_t)n * 4, hipMemcpyDeviceToHost, cx.stream));
    RS_TRY(hipMemcpyAsync(out.delta_ims_model, d_ims.p, (size_t)n * 4, hipMemcpyDeviceToHost, cx.stream));
    RS_TRY(hipGetLastError());
    RS_TRY(hipStreamSynchronize(cx.stream));
    if (out.alignments) std::copy(al.begin(), al.end(), out.alignments);
    float ms = 0.0f;
    RS_TRY(hipEventElapsedTime(&ms, ev.start, ev.stop));
    out.device_ms = ms;
    return true;
}

}  // namespace

// entry point used by capi.hip; returns a SAGE_HIP_* status, message in `err`
int rescore_on_device(int device, const SageRescoreInput& in, SageRescoreOutput& out, std::string& err) {
    Ctx cx;
    if (hipSetDevice(device) != hipSuccess) {
        err = "sage_hip_rescore: hipSetDevice failed";
        return SAGE_HIP_ERR_NO_DEVICE;
    }
    if (!cx.check(hipStreamCreateWithFlags(&cx.stream, hipStreamNonBlocking), "hipStreamCreate")) {
        err = cx.err;
        return cx.code;
    }
    scratch_begin(device, cx.stream);
    const bool ok = rescore_impl(cx, in, out);
    (void)hipStreamSynchronize(cx.stream);
    (void)hipStreamDestroy(cx.stream);
    if (!ok) {
        err = cx.err;
        return cx.code ? cx.code : SAGE_HIP_ERR_HIP;
    }
    return SAGE_HIP_OK;
}

}  // namespace sagehip

namespace sagehip {
int predict_rt_on_device(int device, const SageRtInput& in, SageRtOutput& out, std::string& err) {
    Ctx cx;
    if (hipSetDevice(device) != hipSuccess) {
        err = "sage_hip_predict_rt: hipSetDevice failed";
        return SAGE_HIP_ERR_NO_DEVICE;
    }
    if (!cx.check(hipStreamCreateWithFlags(&cx.stream, hipStreamNonBlocking), "hipStreamCreate")) {
        err = cx.err;
        return cx.code;
    }
    scratch_begin(device, cx.stream);
    const bool ok = predict_rt_impl(cx, in, out);
    (void)hipStreamSynchronize(cx.stream);
    (void)hipStreamDestroy(cx.stream);
    if (!ok) {
        err = cx.err;
        return cx.code ? cx.code : SAGE_HIP_ERR_HIP;
    }
    return SAGE_HIP_OK;
}
}  // namespace sagehip
