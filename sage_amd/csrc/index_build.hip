// index_build.hip — Parameters::build_from_peptides (database.rs:265-346) on the device.
//
// Given the mass-sorted, deduplicated peptide list (the string-heavy part of the build — digestion, modification,
// decoys, reorder_peptides — stays on the host), everything the search kernels read is generated in HBM:
//   * the complete ion table (IonSeries of every peptide and configured kind, ion_series.rs:36-85, same f32 running sum);
//   * the peptide-major fragment list: the entries the reference stores (b3.., y..3 under the default min_ion_index,
//     database.rs:281-292), grouped by peptide;
//   * the tile-major list: the same entries ordered by (tile, m/z, peptide) with ONE radix sort of 64-bit keys — the key
//     is `tile | f32 order key | peptide-in-tile`, 32 - s + 32 + s bits, so the sorted keys ARE the entries;
//   * the per-tile position table (one binary search per cell).
// The reference's global m/z sort + bucketing (database.rs:301-346) is never materialised: the device layouts hold the
// same set of (peptide, m/z) entries, which is all the matching predicate sees.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "device_types.h"

using namespace sagecore;

namespace sagehip {

namespace {

// mass.rs:64-68 (A..Z); zeros for B, J, X, Z
__constant__ float kResidueDev[26] = {71.03711f,  0.0f,       103.00919f, 115.02694f, 129.04259f, 147.0684f,  57.02146f,
                                      137.05891f, 113.08406f, 0.0f,       128.09496f, 113.08406f, 131.0405f,  114.04293f,
                                      237.14774f, 97.05276f,  128.05858f, 156.1011f,  87.03203f,  101.04768f, 150.95363f,
                                      99.06841f,  186.07932f, 0.0f,       163.06332f, 0.0f};

// one thread per (peptide, ion kind): IonSeries::new(peptide, kind) (ion_series.rs:36-85) into the ion table, and the
// stored subset (database.rs:281-292) into the peptide-major fragment list
__global__ __launch_bounds__(256) void ion_kernel(uint64_t np, uint32_t nk, const uint8_t* __restrict__ kinds,
                                                  const uint64_t* __restrict__ seq_off, const uint8_t* __restrict__ seq,
                                                  const float* __restrict__ mods, const float* __restrict__ nterm,
                                                  const float* __restrict__ mono, uint64_t min_ion_index,
                                                  const uint64_t* __restrict__ ion_off, const uint64_t* __restrict__ pm_off,
                                                  float* __restrict__ ions, SageTheoretical* __restrict__ pm_frag) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= np * nk) return;
    const uint64_t p = gid / nk;
    const uint32_t k = (uint32_t)(gid - p * nk);
    const uint8_t kind = kinds[k];
    const uint64_t s0 = seq_off[p];
    const uint64_t len = seq_off[p + 1] - s0, lm1 = len ? len - 1 : 0;
    const float C = 12.0f, O = 15.994914f, H = 1.007825f, PRO = 1.0072764f, N = 14.003074f;
    const float NH3 = N + H * 2.0f + PRO;
    const float ntv = nterm[p];
    const float nt = ntv == ntv ? ntv : 0.0f;  // NaN == None
    const float m = mono[p];
    float cum;
    switch (kind) {
        case SAGE_ION_A: cum = nt - (C + O); break;
        case SAGE_ION_B: cum = nt; break;
        case SAGE_ION_C: cum = nt + NH3; break;
        case SAGE_ION_X: cum = m - nt + (C + O - NH3 + N + H); break;
        case SAGE_ION_Y: cum = m - nt; break;
        default: cum = m - nt - NH3; break;
    }
    const bool forward = kind <= SAGE_ION_C;
    const uint64_t kept = lm1 > min_ion_index ? lm1 - min_ion_index : 0;
    float* out = ions + ion_off[p] + (uint64_t)k * lm1;
    SageTheoretical* fr = pm_frag + pm_off[p] + (uint64_t)k * kept;
    uint64_t w = 0;
    for (uint64_t i = 0; i < lm1; i++) {
        const uint8_t aa = seq[s0 + i];
        const float r = (aa >= 'A' && aa <= 'Z') ? kResidueDev[aa - 'A'] : 0.0f;
        const float step = r + mods[s0 + i];
        cum += forward ? step : -step;
        out[i] = cum;
        const bool keep = forward ? (i + 1) > min_ion_index : (lm1 - i) > min_ion_index;
        if (keep) fr[w++] = SageTheoretical{(uint32_t)p, cum};
    }
}

__global__ __launch_bounds__(256) void encode_kernel(uint64_t nf, uint32_t tile_shift, const SageTheoretical* __restrict__ pm,
                                                     uint64_t* __restrict__ keys) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf) return;
    const SageTheoretical e = pm[i];
    const uint32_t mzkey = (uint32_t)order_key(e.fragment_mz) ^ 0x80000000u;  // unsigned order == f32::total_cmp
    const uint64_t tile = e.peptide_index >> tile_shift, low = e.peptide_index & ((1u << tile_shift) - 1u);
    keys[i] = (tile << (32 + tile_shift)) | ((uint64_t)mzkey << tile_shift) | low;
}

__global__ __launch_bounds__(256) void decode_kernel(uint64_t nf, uint32_t tile_shift, const uint64_t* __restrict__ keys,
                                                     SageTheoretical* __restrict__ tm) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf + 2) return;
    if (i >= nf) {  // padding read (never matched) by the 16-byte loads of the search kernels
        tm[i] = SageTheoretical{0xFFFFFFFFu, 0.0f};
        return;
    }
    const uint64_t k = keys[i];
    const uint32_t pep = (uint32_t)((k >> (32 + tile_shift)) << tile_shift) | (uint32_t)(k & ((1u << tile_shift) - 1u));
    int32_t o = (int32_t)((uint32_t)(k >> tile_shift) ^ 0x80000000u);
    o ^= (int32_t)(((uint32_t)(o >> 31)) >> 1);  // inverse of order_key
    tm[i] = SageTheoretical{pep, __int_as_float(o)};
}

// the position table of a tile-major copy: entry (t, c) = first position of tile t whose m/z is >= c / scale (c == 0: the tile's
// start, c == last: its end).  Row-major, lut[t][c] — the small tiles of the narrow kernel, where a window meets one or two tiles —
// or TRANSPOSED, lut[c][t] — the large tiles of the open-search kernel, where ONE window is looked up in the ~100 consecutive
// tiles of a precursor window: the words of consecutive tiles then share a cache line (32 tiles per 128 bytes) instead of
// costing a line each.
template <int LAYOUT>  // device_types.h: TM_LUT_LAYOUT (0 row-major, 1 transposed, 2 quads of tiles)
__global__ __launch_bounds__(256) void lut_kernel(uint32_t n_tiles, uint32_t lut_stride, float lut_scale,
                                                  const uint64_t* __restrict__ tile_off, const SageTheoretical* __restrict__ tm,
                                                  uint32_t* __restrict__ lut) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t rows = LAYOUT == 2 ? (n_tiles + 3u) & ~3u : n_tiles;
    if (gid >= (uint64_t)rows * lut_stride) return;
    uint32_t t, c;
    if (LAYOUT == 2) {
        t = (uint32_t)((gid >> 2) / lut_stride) * 4u + (uint32_t)(gid & 3u);
        c = (uint32_t)((gid >> 2) % lut_stride);
    } else if (LAYOUT == 1) {
        c = (uint32_t)(gid / n_tiles);
        t = (uint32_t)(gid - (uint64_t)c * n_tiles);
    } else {
        t = (uint32_t)(gid / lut_stride);
        c = (uint32_t)(gid - (uint64_t)t * lut_stride);
    }
    static_assert(sizeof(SageTheoretical) == 8, "m/z is every second float of the entry array");
    // (a row beyond the last tile — padding of the last quad — is an empty tile at the end of the array)
    lut[gid] = t < n_tiles ? sagecore::lut_entry(&tm[0].fragment_mz, 2, tile_off[t], tile_off[t + 1], c, lut_stride, lut_scale)
                           : (uint32_t)tile_off[n_tiles];
}

__global__ __launch_bounds__(256) void maxmz_kernel(uint64_t nf, const SageTheoretical* __restrict__ pm, uint32_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float v = 0.0f;
    if (i < nf) {
        const float m = pm[i].fragment_mz;
        if (m == m && m < 3.0e38f && m > 0.0f) v = m;
    }
    // positive floats order like their bit patterns
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    if ((threadIdx.x & 63u) == 0 && v > 0.0f) atomicMax(out + (blockIdx.x & 255u), __float_as_uint(v));  // (256 slots: no hot address)
}

// smallest and largest |ion| of the rescoring table, as bit patterns (non-negative floats order like their bits; a NaN sorts above
// +inf, so it shows in the maximum): out[0] = min, out[1] = max
__global__ __launch_bounds__(256) void ion_range_kernel(uint64_t n, const float* __restrict__ ions, uint32_t* __restrict__ out) {
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t a = __float_as_uint(ions[i]) & 0x7FFFFFFFu;
        lo = a < lo ? a : lo;
        hi = a > hi ? a : hi;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t l = (uint32_t)__shfl_xor((int)lo, off, 64), h = (uint32_t)__shfl_xor((int)hi, off, 64);
        lo = l < lo ? l : lo;
        hi = h > hi ? h : hi;
    }
    if ((threadIdx.x & 63u) == 0) {
        atomicMin(out, lo);
        atomicMax(out + 1, hi);
    }
}

}  // namespace

#define BUILD_TRY(expr)                  \
    do {                                 \
        hipError_t _e = (expr);          \
        if (_e != hipSuccess) return _e; \
    } while (0)

// Both functions return a hipError_t; all pointers are device pointers.

// The range of |ion| over the n entries of the rescoring table (bit patterns of f32; n == 0: {0xFFFFFFFF, 0}).
int ion_abs_range_on_device(const float* d_ions, uint64_t n, uint32_t* lo_bits, uint32_t* hi_bits) {
    uint32_t* d_out = nullptr;
    uint32_t h[2] = {0xFFFFFFFFu, 0u};
    BUILD_TRY(hipMalloc(&d_out, sizeof(h)));
    hipError_t e = hipMemcpy(d_out, h, sizeof(h), hipMemcpyHostToDevice);
    if (e == hipSuccess && n) {
        const uint32_t grid = (uint32_t)std::min<uint64_t>((n + 255) / 256, 4096);
        hipLaunchKernelGGL(ion_range_kernel, dim3(grid), dim3(256), 0, 0, n, d_ions, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    *lo_bits = h[0];
    *hi_bits = h[1];
    return e;
}

// IonSeries of every peptide and kind -> d_ions; the stored subset -> d_pm_frag (peptide-major).  Buffers are the caller's.
int generate_fragments_on_device(uint64_t np, uint32_t nk, const uint8_t* d_kinds, const uint64_t* d_seq_off, const uint8_t* d_seq,
                                 const float* d_mods, const float* d_nterm, const float* d_mono, uint64_t min_ion_index,
                                 const uint64_t* d_ion_off, const uint64_t* d_pm_off, float* d_ions, SageTheoretical* d_pm_frag,
                                 void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const uint64_t nthreads = np * nk;
    if (nthreads) hipLaunchKernelGGL(ion_kernel, dim3((uint32_t)((nthreads + 255) / 256)), dim3(256), 0, stream, np, nk, d_kinds,
                                     d_seq_off, d_seq, d_mods, d_nterm, d_mono, min_ion_index, d_ion_off, d_pm_off, d_ions, d_pm_frag);
    return (int)hipGetLastError();
}

// The precursor-window search key's own table: lut[b] = partition_point(order_key(pep_mono[i]) < order_key(b * w)), b = 0..bins,
// w = 1 / inv_w a power of two — a precursor window's two partition points (IndexedDatabase::query, database.rs:402-425) are then
// two scalar reads of this table and one wave-wide read of pep_mono each, instead of a five-level search (kernels.hip:
// query_window).  bins * w > the largest mass, so lut[bins] == np.  bins == 0: no table (an empty database, a top mass that is
// negative or not finite).
__global__ __launch_bounds__(256) void pepmass_lut_kernel(const float* __restrict__ pep_mono, uint32_t np, uint32_t bins, float inv_w,
                                                          uint32_t* __restrict__ lut) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > bins) return;
    const int32_t edge = sagecore::order_key((float)b / inv_w);  // (b < 2^23, w a power of two: exact)
    uint32_t lo = 0, hi = np;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (sagecore::order_key(pep_mono[mid]) < edge) lo = mid + 1; else hi = mid;
    }
    lut[b] = lo;
}
int build_peptide_mass_lut(const float* d_pep_mono, uint32_t np, float top_mass, uint32_t** d_lut_out, uint32_t* bins_out, float* inv_w_out,
                           void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    *d_lut_out = nullptr;
    *bins_out = 0;
    *inv_w_out = 0.0f;
    if (np == 0 || !(top_mass >= 0.0f) || !(top_mass < 1.0e30f)) return (int)hipSuccess;
    float inv_w = 128.0f;  // 1/128 Da: ~4 peptides of a human tryptic database per bin on average
    while ((double)top_mass * inv_w + 2.0 > 4194304.0) inv_w *= 0.5f;
    const uint32_t bins = (uint32_t)((double)top_mass * inv_w) + 1u;
    uint32_t* d_lut = nullptr;
    BUILD_TRY(hipMalloc((void**)&d_lut, ((size_t)bins + 1) * 4));
    hipLaunchKernelGGL(pepmass_lut_kernel, dim3((bins + 1 + 255) / 256), dim3(256), 0, stream, d_pep_mono, np, bins, inv_w, d_lut);
    BUILD_TRY(hipGetLastError());
    BUILD_TRY(hipStreamSynchronize(stream));
    *d_lut_out = d_lut;
    *bins_out = bins;
    *inv_w_out = inv_w;
    return (int)hipSuccess;
}

// A tile-major copy of the peptide-major list for tiles of 2^tile_shift peptides (d_tile_off: [n_tiles + 1] fragment
// offsets of the tile boundaries) + its position table at `lut_scale` cells per Da.  d_tm_frag ([nf + 2]) is the caller's,
// the table is allocated here (its width depends on the largest fragment m/z).
int build_tile_copy_on_device(const SageTheoretical* d_pm_frag, uint64_t nf, uint32_t tile_shift, uint32_t n_tiles,
                              const uint64_t* d_tile_off, float lut_scale, SageTheoretical* d_tm_frag, uint32_t** d_lut_out,
                              uint32_t* lut_stride_out, void* stream_, int layout) {
    hipStream_t stream = (hipStream_t)stream_;
    // largest finite fragment m/z -> table width
    uint32_t* d_max = nullptr;
    BUILD_TRY(hipMalloc((void**)&d_max, 256 * 4));
    BUILD_TRY(hipMemsetAsync(d_max, 0, 256 * 4, stream));
    if (nf) hipLaunchKernelGGL(maxmz_kernel, dim3((uint32_t)((nf + 255) / 256)), dim3(256), 0, stream, nf, d_pm_frag, d_max);
    uint32_t max_slots[256] = {};
    BUILD_TRY(hipMemcpyAsync(max_slots, d_max, 256 * 4, hipMemcpyDeviceToHost, stream));
    BUILD_TRY(hipStreamSynchronize(stream));
    (void)hipFree(d_max);
    uint32_t max_bits = 0;  // positive floats order like their bit patterns
    for (uint32_t v : max_slots) max_bits = v > max_bits ? v : max_bits;
    float max_mz;
    memcpy(&max_mz, &max_bits, 4);
    const double cells = ceil((double)max_mz * lut_scale) + 3.0;
    const uint32_t lut_stride = (uint32_t)(cells < 64.0e6 ? cells : 64.0e6);
    if ((double)n_tiles * lut_stride > 4.0e9) return (int)hipErrorInvalidValue;
    // (tile, m/z, peptide) order: one radix sort of 64-bit keys
    uint64_t *k_in = nullptr, *k_out = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    BUILD_TRY(hipMalloc((void**)&k_in, (nf ? nf : 1) * 8));
    BUILD_TRY(hipMalloc((void**)&k_out, (nf ? nf : 1) * 8));
    if (nf) {
        hipLaunchKernelGGL(encode_kernel, dim3((uint32_t)((nf + 255) / 256)), dim3(256), 0, stream, nf, tile_shift, d_pm_frag, k_in);
        BUILD_TRY(rocprim::radix_sort_keys(nullptr, tmp_bytes, k_in, k_out, nf, 0, 64, stream));
        BUILD_TRY(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 1));
        BUILD_TRY(rocprim::radix_sort_keys(tmp, tmp_bytes, k_in, k_out, nf, 0, 64, stream));
    }
    hipLaunchKernelGGL(decode_kernel, dim3((uint32_t)((nf + 2 + 255) / 256)), dim3(256), 0, stream, nf, tile_shift, k_out, d_tm_frag);
    BUILD_TRY(hipGetLastError());
    uint32_t* d_lut = nullptr;
    const uint64_t lut_n = (uint64_t)(layout == 2 ? (n_tiles + 3u) & ~3u : n_tiles) * lut_stride;
    BUILD_TRY(hipMalloc((void**)&d_lut, (lut_n ? lut_n : 1) * 4));
    const dim3 lut_grid((uint32_t)((lut_n + 255) / 256));
    if (layout == 2)
        hipLaunchKernelGGL(lut_kernel<2>, lut_grid, dim3(256), 0, stream, n_tiles, lut_stride, lut_scale, d_tile_off, d_tm_frag, d_lut);
    else if (layout == 1)
        hipLaunchKernelGGL(lut_kernel<1>, lut_grid, dim3(256), 0, stream, n_tiles, lut_stride, lut_scale, d_tile_off, d_tm_frag, d_lut);
    else
        hipLaunchKernelGGL(lut_kernel<0>, lut_grid, dim3(256), 0, stream, n_tiles, lut_stride, lut_scale, d_tile_off, d_tm_frag, d_lut);
    BUILD_TRY(hipGetLastError());
    BUILD_TRY(hipStreamSynchronize(stream));
    (void)hipFree(k_in);
    (void)hipFree(k_out);
    if (tmp) (void)hipFree(tmp);
    *d_lut_out = d_lut;
    *lut_stride_out = lut_stride;
    return (int)hipSuccess;
}


// The peptide-major fragment list out of a tile-major copy (either one: they hold the same entries): the tile copy's own sort with
// tiles of ONE peptide — the key is (peptide, m/z).  The order inside a peptide is by m/z, not by (kind, ion index) as generated; the
// stream variant of the preliminary kernels, the list's only reader, counts matches and does not care.  d_pm_frag: [nf + 2].
int rebuild_peptide_major_on_device(const SageTheoretical* d_tm_frag, uint64_t nf, SageTheoretical* d_pm_frag, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    uint64_t *k_in = nullptr, *k_out = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    BUILD_TRY(hipMalloc((void**)&k_in, (nf ? nf : 1) * 8));
    BUILD_TRY(hipMalloc((void**)&k_out, (nf ? nf : 1) * 8));
    if (nf) {
        hipLaunchKernelGGL(encode_kernel, dim3((uint32_t)((nf + 255) / 256)), dim3(256), 0, stream, nf, 0u, d_tm_frag, k_in);
        BUILD_TRY(rocprim::radix_sort_keys(nullptr, tmp_bytes, k_in, k_out, nf, 0, 64, stream));
        BUILD_TRY(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 1));
        BUILD_TRY(rocprim::radix_sort_keys(tmp, tmp_bytes, k_in, k_out, nf, 0, 64, stream));
    }
    hipLaunchKernelGGL(decode_kernel, dim3((uint32_t)((nf + 2 + 255) / 256)), dim3(256), 0, stream, nf, 0u, k_out, d_pm_frag);
    BUILD_TRY(hipGetLastError());
    BUILD_TRY(hipStreamSynchronize(stream));
    (void)hipFree(k_in);
    (void)hipFree(k_out);
    if (tmp) (void)hipFree(tmp);
    return (int)hipSuccess;
}

// ---- the succinct form of a row-major position table (core.h: LutWord) -----------------------------------------------------------
namespace {
// one thread per (tile, word): the occupancy bits of its 32 cells — cell c holds an entry iff lut[c + 1] != lut[c] (the table
// ascends; its last entry, c == stride - 1, is the tile's end and has no cell behind it) — and their count; the last word of a
// tile counts one more: the slot of the tile's end in `pos`
__global__ __launch_bounds__(256) void lut_bits_kernel(const uint32_t* __restrict__ lut, uint32_t n_tiles, uint32_t stride, uint32_t words,
                                                       sagecore::LutWord* __restrict__ l1, uint32_t* __restrict__ counts) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (uint64_t)n_tiles * words) return;
    const uint32_t t = (uint32_t)(gid / words), w = (uint32_t)(gid - (uint64_t)t * words);
    const uint32_t* __restrict__ row = lut + (size_t)t * stride;
    uint32_t bits = 0;
    for (uint32_t b = 0; b < 32; b++) {
        const uint32_t c = w * 32 + b;
        if (c + 1 < stride && row[c + 1] != row[c]) bits |= 1u << b;
    }
    l1[gid].bits = bits;
    counts[gid] = (uint32_t)__popc(bits) + (w + 1 == words ? 1u : 0u);
}
// ... then, with the exclusive prefix of the counts as every word's rank: the run starts of its non-empty cells, and behind a
// tile's last word the tile's end
__global__ __launch_bounds__(256) void lut_fill_kernel(const uint32_t* __restrict__ lut, uint32_t n_tiles, uint32_t stride, uint32_t words,
                                                       const uint32_t* __restrict__ ranks, sagecore::LutWord* __restrict__ l1,
                                                       uint32_t* __restrict__ pos) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (uint64_t)n_tiles * words) return;
    const uint32_t t = (uint32_t)(gid / words), w = (uint32_t)(gid - (uint64_t)t * words);
    const uint32_t* __restrict__ row = lut + (size_t)t * stride;
    uint32_t r = ranks[gid];
    l1[gid].rank = r;
    uint32_t bits = l1[gid].bits;
    while (bits) {
        const uint32_t b = (uint32_t)__ffs((int)bits) - 1;
        bits &= bits - 1;
        pos[r++] = row[w * 32 + b];
    }
    if (w + 1 == words) pos[r] = row[stride - 1];
}
}  // namespace

int build_succinct_lut_on_device(const uint32_t* d_lut, uint32_t n_tiles, uint32_t lut_stride, sagecore::LutWord** d_l1_out, uint32_t** d_pos_out,
                                 uint32_t* words_out, uint64_t* n_pos_out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const uint32_t words = (lut_stride + 31) / 32;
    const uint64_t n = (uint64_t)n_tiles * words;
    sagecore::LutWord* d_l1 = nullptr;
    uint32_t *d_counts = nullptr, *d_ranks = nullptr, *d_pos = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    BUILD_TRY(hipMalloc((void**)&d_l1, (n ? n : 1) * sizeof(sagecore::LutWord)));
    BUILD_TRY(hipMalloc((void**)&d_counts, (n ? n : 1) * 4));
    BUILD_TRY(hipMalloc((void**)&d_ranks, (n ? n : 1) * 4));
    uint64_t total = 0;
    if (n) {
        const dim3 grid((uint32_t)((n + 255) / 256));
        hipLaunchKernelGGL(lut_bits_kernel, grid, dim3(256), 0, stream, d_lut, n_tiles, lut_stride, words, d_l1, d_counts);
        BUILD_TRY(hipGetLastError());
        BUILD_TRY(rocprim::exclusive_scan(nullptr, tmp_bytes, d_counts, d_ranks, 0u, n, rocprim::plus<uint32_t>(), stream));
        BUILD_TRY(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 1));
        BUILD_TRY(rocprim::exclusive_scan(tmp, tmp_bytes, d_counts, d_ranks, 0u, n, rocprim::plus<uint32_t>(), stream));
        uint32_t last_rank = 0, last_count = 0;
        BUILD_TRY(hipMemcpyAsync(&last_rank, d_ranks + (n - 1), 4, hipMemcpyDeviceToHost, stream));
        BUILD_TRY(hipMemcpyAsync(&last_count, d_counts + (n - 1), 4, hipMemcpyDeviceToHost, stream));
        BUILD_TRY(hipStreamSynchronize(stream));
        total = (uint64_t)last_rank + last_count;  // (<= entries + tiles < 2^32: capi.hip refuses more fragments)
        BUILD_TRY(hipMalloc((void**)&d_pos, (total ? total : 1) * 4));
        hipLaunchKernelGGL(lut_fill_kernel, grid, dim3(256), 0, stream, d_lut, n_tiles, lut_stride, words, d_ranks, d_l1, d_pos);
        BUILD_TRY(hipGetLastError());
        BUILD_TRY(hipStreamSynchronize(stream));
    } else {
        BUILD_TRY(hipMalloc((void**)&d_pos, 4));
    }
    (void)hipFree(d_counts);
    (void)hipFree(d_ranks);
    if (tmp) (void)hipFree(tmp);
    *d_l1_out = d_l1;
    *d_pos_out = d_pos;
    *words_out = words;
    *n_pos_out = total;
    return (int)hipSuccess;
}

// ---- launch schedule of a spectrum batch -------------------------------------------------------------------------------
// Spectra are scored in ascending order of their neutral precursor mass, so that wavefronts resident together read
// overlapping ranges of the index and of the ion table (outputs keep input order).  One stable 32-bit radix sort on the
// device instead of a host sort: the schedule of a batch is ready a few microseconds after its upload.
namespace {
__global__ __launch_bounds__(256) void schedule_keys_kernel(uint32_t n, const float* __restrict__ precursor_mz,
                                                            const uint8_t* __restrict__ charge, uint32_t min_charge,
                                                            uint32_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t z = charge[i] ? charge[i] : min_charge;
    const float m = (precursor_mz[i] - sagecore::PROTON) * (float)z;  // scoring.rs:420 (first charge when unknown)
    const uint32_t b = __float_as_uint(m);
    keys[i] = b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);  // ascending u32 == ascending f32 (total order)
    idx[i] = i;
}
// DevBatchView::sched: the per-spectrum words a scoring block starts from, gathered once per upload into schedule order
__global__ __launch_bounds__(256) void schedule_records_kernel(uint32_t n, const uint32_t* __restrict__ order, const uint64_t* __restrict__ peak_off,
                                                               const float* __restrict__ precursor_mz, const uint8_t* __restrict__ charge,
                                                               const float* __restrict__ iso_lo, const float* __restrict__ iso_hi,
                                                               uint4* __restrict__ sched) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const uint32_t i = order[k];
    const uint64_t p0 = peak_off[i];
    const uint32_t nan_bits = 0x7FC00000u;
    sched[2 * (size_t)k] = make_uint4(i, (uint32_t)(peak_off[i + 1] - p0), (uint32_t)p0, (uint32_t)(p0 >> 32));
    sched[2 * (size_t)k + 1] = make_uint4(charge[i], __float_as_uint(precursor_mz[i]), iso_lo && iso_hi ? __float_as_uint(iso_lo[i]) : nan_bits,
                                          iso_lo && iso_hi ? __float_as_uint(iso_hi[i]) : nan_bits);
}
}  // namespace

size_t schedule_temp_bytes(uint32_t n) {
    size_t bytes = 0;
    uint32_t* p = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, p, p, p, p, n ? n : 1, 0, 32, (hipStream_t) nullptr);
    return bytes ? bytes : 16;
}

int schedule_on_device(uint32_t n, const float* d_precursor_mz, const uint8_t* d_charge, uint32_t min_charge, uint32_t* d_keys_a,
                       uint32_t* d_keys_b, uint32_t* d_idx, uint32_t* d_order, void* d_temp, size_t temp_bytes, void* stream_) {
    if (n == 0) return (int)hipSuccess;
    hipStream_t stream = (hipStream_t)stream_;
    hipLaunchKernelGGL(schedule_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, d_precursor_mz, d_charge, min_charge,
                       d_keys_a, d_idx);
    size_t bytes = temp_bytes;
    return (int)rocprim::radix_sort_pairs(d_temp, bytes, d_keys_a, d_keys_b, d_idx, d_order, n, 0, 32, stream);
}

void schedule_records_on_device(uint32_t n, const uint32_t* d_order, const uint64_t* d_peak_off, const float* d_precursor_mz,
                                const uint8_t* d_charge, const float* d_iso_lo, const float* d_iso_hi, uint4* d_sched, void* stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(schedule_records_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, d_order, d_peak_off,
                       d_precursor_mz, d_charge, d_iso_lo, d_iso_hi, d_sched);
}

}  // namespace sagehip
