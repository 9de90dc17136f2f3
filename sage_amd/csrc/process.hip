// process.hip — SpectrumProcessor::process for MS2 spectra on the device (spectrum.rs:179-227, 279-412): the step
// immediately before the search-and-score path.  Raw centroided peaks go in, the
// ProcessedSpectrum arrays (masses ascending, intensities, total ion current) come out IN HBM, where the scoring kernels
// read them — no host round trip between the mzML decode and Scorer::score.
//
// One 64-lane wavefront owns one spectrum; the spectrum lives in LDS — or, for the rare spectrum with more raw peaks than a
// reasonable LDS share holds (PROCESS_LDS_PEAKS), in a slice of a global-memory workspace: the same code through generic
// pointers (process_kernel<true>), slower per access but without a limit on the peak count, and without making every other
// wavefront of the batch carry that spectrum's LDS footprint.
//   deisotope (spectrum.rs:179-227) is a sequential two-pointer loop whose `+=` chain runs from high to low m/z and whose
//     quirks (the `j == 0` break, the "already part of an envelope of another charge" skip) are observable: lane 0
//     replays it verbatim out of LDS;
//   the (intensity desc, m/z asc) sort, the top-N cut, the stable mass sort are bitonic sorts of a permutation in LDS
//     with the original index as the last key (== the stable order the CPU restatement uses);
//   the non-deisotoping branch selects with bounded_min_heapify (heap.rs:7-28) — its layout leaks into the order of
//     equal-mass peaks — so lane 0 replays the heap as well;
//   total_ion_current is summed by one lane in ascending-mass order (f32, spectrum.rs:397).
// Every f32 expression keeps the reference's operation order (compile with -ffp-contract=off).
#include <hip/hip_runtime.h>

#include "device_types.h"

using namespace sagecore;

namespace sagehip {

namespace {

constexpr uint32_t WAVE = 64;

struct ProcLds {
    float* mz;        // [rcap] raw m/z
    float* inten;     // [rcap] raw intensity
    float* acc;       // [rcap] envelope-summed intensity (Deisotoped.intensity) / Peak.intensity
    float* mass;      // [rcap] Peak.mass
    uint32_t* perm;   // [rpow2] permutation being sorted
    uint8_t* charge;  // [rcap] 0 == None
    uint8_t* child;   // [rcap] 1 == envelope.is_some()
};

__device__ __forceinline__ void wsync() { __syncthreads(); }  // (one wavefront per block)

// bitonic sort of perm[0..npow2) by `before(a, b)` (a strict total order on element ids; ids >= n sort last)
template <class Before>
__device__ __forceinline__ void bitonic_sort(uint32_t* perm, uint32_t npow2, Before before) {
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t k = 2; k <= npow2; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = lane; t < npow2 / 2; t += WAVE) {
                const uint32_t i = 2 * t - (t & (j - 1));  // lower index of the pair with distance j
                const uint32_t l = i + j;
                const uint32_t a = perm[i], b = perm[l];
                const bool up = (i & k) == 0;
                if (up ? before(b, a) : before(a, b)) {
                    perm[i] = b;
                    perm[l] = a;
                }
            }
            wsync();
        }
    }
}

// BIG == false: block b owns spectrum b, skipped when it has more than rcap raw peaks (the other instance's); working arrays in
// LDS.  BIG == true: block b owns spectrum big_list[b]; working arrays in workspace + b * ws_stride (rcap / rpow2 are the
// capacity of a slice).
template <bool BIG>
__global__ __launch_bounds__(64) void process_kernel(uint32_t n_spectra, const uint64_t* __restrict__ raw_off,
                                                     const float* __restrict__ raw_mz, const float* __restrict__ raw_int,
                                                     const uint8_t* __restrict__ precursor_charge, uint32_t take_top_n,
                                                     uint32_t deisotope, float min_deisotope_mz, uint32_t rcap, uint32_t rpow2,
                                                     uint32_t stride, float* __restrict__ out_mass, float* __restrict__ out_int,
                                                     float* __restrict__ out_tic, uint32_t* __restrict__ out_count,
                                                     const uint32_t* __restrict__ big_list, unsigned char* workspace, size_t ws_stride) {
    extern __shared__ __align__(16) unsigned char lds[];
    const uint32_t lane = threadIdx.x & 63u;
    if (blockIdx.x >= n_spectra) return;
    const uint32_t spec = BIG ? big_list[blockIdx.x] : blockIdx.x;
    if (!BIG && raw_off[spec + 1] - raw_off[spec] > rcap) return;
    unsigned char* const smem = BIG ? workspace + (size_t)blockIdx.x * ws_stride : lds;
    ProcLds L;
    L.mz = (float*)smem;
    L.inten = L.mz + rcap;
    L.acc = L.inten + rcap;
    L.mass = L.acc + rcap;
    L.perm = (uint32_t*)(L.mass + rcap);
    L.charge = (uint8_t*)(L.perm + rpow2);
    L.child = L.charge + rcap;
    const uint64_t r0 = raw_off[spec];
    const uint32_t n = (uint32_t)(raw_off[spec + 1] - r0);
    for (uint32_t i = lane; i < n; i += WAVE) {
        const float m = raw_mz[r0 + i], it = raw_int[r0 + i];
        L.mz[i] = m;
        L.inten[i] = it;
        L.acc[i] = it;
        L.charge[i] = 0;
        L.child[i] = 0;
    }
    wsync();
    uint32_t npow2 = 1;
    while (npow2 < n) npow2 <<= 1;
    uint32_t kept = 0;  // peaks that survive, in perm[0..kept)
    if (deisotope) {
        if (lane == 0 && n) {  // spectrum.rs:198-225, verbatim
            const uint32_t zraw = precursor_charge[spec];
            const uint32_t max_charge = zraw ? zraw : 3;  // spectrum.rs:289-293
            const float ppm = 10.0f;
            for (uint32_t i = n; i-- > 0;) {
                const float mi = L.mz[i];
                const float tol = ppm * mi / 1000000.0f;  // Tolerance::ppm_to_delta_mass
                uint32_t j = i ? i - 1 : 0;
                while (mi - L.mz[j] <= NEUTRON + tol && L.mz[j] >= min_deisotope_mz) {
                    const float delta = mi - L.mz[j];
                    for (uint32_t z = 1; z <= max_charge; z++) {
                        const float iso = NEUTRON / (float)z;
                        if (__builtin_fabsf(delta - iso) <= tol && L.inten[i] < L.inten[j]) {
                            if (L.charge[i] && L.charge[i] != z) continue;  // already in an envelope of another charge
                            L.acc[j] += L.acc[i];
                            L.charge[j] = (uint8_t)z;
                            L.charge[i] = (uint8_t)z;
                            L.child[i] = 1;
                        }
                    }
                    j = j ? j - 1 : 0;
                    if (j == 0) break;
                }
            }
        }
        wsync();
        // sort_unstable_by(intensity desc, then m/z asc) (spectrum.rs:303-307); equal keys keep index order; peaks that
        // are part of an envelope are filtered afterwards (:310), i.e. they simply never count towards take_top_n
        for (uint32_t i = lane; i < npow2; i += WAVE) L.perm[i] = i;
        wsync();
        auto before = [&](uint32_t a, uint32_t b) -> bool {
            if (a >= n || b >= n) return a < n;  // padding last (and a < b among padding does not matter)
            const bool ca = L.child[a] != 0, cb = L.child[b] != 0;
            if (ca != cb) return cb;  // envelope members after every kept peak
            const int32_t ia = order_key(L.acc[a]), ib = order_key(L.acc[b]);
            if (ia != ib) return ia > ib;
            const int32_t ma = order_key(L.mz[a]), mb = order_key(L.mz[b]);
            if (ma != mb) return ma < mb;
            return a < b;
        };
        bitonic_sort(L.perm, npow2, before);
        uint32_t nonchild = 0;
        for (uint32_t i = lane; i < n; i += WAVE) nonchild += L.child[i] ? 0u : 1u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nonchild += __shfl_xor(nonchild, off, 64);
        kept = nonchild < take_top_n ? nonchild : take_top_n;
        for (uint32_t i = lane; i < kept; i += WAVE) {
            const uint32_t p = L.perm[i];
            const uint32_t z = L.charge[p] ? L.charge[p] : 1;
            L.mass[p] = (L.mz[p] - PROTON) * (float)z;  // MH* -> M (spectrum.rs:314)
        }
        wsync();
    } else {
        for (uint32_t i = lane; i < n; i += WAVE) L.mass[i] = (L.mz[i] - PROTON) * 1.0f;  // spectrum.rs:328
        for (uint32_t i = lane; i < npow2; i += WAVE) L.perm[i] = i;
        wsync();
        kept = n < take_top_n ? n : take_top_n;
        if (lane == 0 && n > take_top_n) {  // bounded_min_heapify(&mut peaks, k) + truncate (spectrum.rs:332-333)
            const uint32_t k = take_top_n;
            auto less = [&](uint32_t a, uint32_t b) -> bool {  // Peak: intensity, then mass (total_cmp)
                const int32_t ia = order_key(L.acc[a]), ib = order_key(L.acc[b]);
                if (ia != ib) return ia < ib;
                return order_key(L.mass[a]) < order_key(L.mass[b]);
            };
            auto sift = [&](uint32_t idx) {
                for (;;) {
                    const uint32_t l = 2 * idx + 1, r = l + 1;
                    uint32_t s = idx;
                    if (l < k && less(L.perm[l], L.perm[s])) s = l;
                    if (r < k && less(L.perm[r], L.perm[s])) s = r;
                    if (s == idx) break;
                    const uint32_t t = L.perm[s];
                    L.perm[s] = L.perm[idx];
                    L.perm[idx] = t;
                    idx = s;
                }
            };
            for (uint32_t i = k / 2; i-- > 0;) sift(i);
            for (uint32_t i = k; i < n; i++) {
                if (less(L.perm[0], L.perm[i])) {
                    const uint32_t t = L.perm[0];
                    L.perm[0] = L.perm[i];
                    L.perm[i] = t;
                    sift(0);
                }
            }
        }
        wsync();
    }
    // peaks.sort_by(mass.total_cmp) (spectrum.rs:391): stable, so equal masses keep the order they have in perm[0..kept)
    // — re-key the kept entries by their position and sort (mass asc, position asc)
    uint32_t kpow2 = 1;
    while (kpow2 < kept) kpow2 <<= 1;
    uint32_t* pos = (uint32_t*)L.inten;  // raw intensities are no longer needed: position of element id in the kept list
    for (uint32_t i = lane; i < kept; i += WAVE) pos[L.perm[i]] = i;
    for (uint32_t i = kept + lane; i < kpow2; i += WAVE) L.perm[i] = 0xFFFFFFFFu;
    wsync();
    auto before_mass = [&](uint32_t a, uint32_t b) -> bool {
        if (a == 0xFFFFFFFFu || b == 0xFFFFFFFFu) return a != 0xFFFFFFFFu;
        const int32_t ma = order_key(L.mass[a]), mb = order_key(L.mass[b]);
        if (ma != mb) return ma < mb;
        return pos[a] < pos[b];
    };
    bitonic_sort(L.perm, kpow2, before_mass);
    float* om = out_mass + (size_t)spec * stride;
    float* oi = out_int + (size_t)spec * stride;
    for (uint32_t i = lane; i < kept; i += WAVE) {
        const uint32_t p = L.perm[i];
        om[i] = L.mass[p];
        oi[i] = L.acc[p];
    }
    if (lane == 0) {
        float tic = 0.0f;  // intensities.iter().sum::<f32>() (spectrum.rs:397)
        for (uint32_t i = 0; i < kept; i++) tic += L.acc[L.perm[i]];
        out_tic[spec] = tic;
        out_count[spec] = kept;
    }
}

// strided -> dense: spectrum i's peaks go to [peak_off[i], peak_off[i + 1]) (zero peaks when below min_peaks)
__global__ __launch_bounds__(64) void compact_kernel(uint32_t n_spectra, const uint64_t* __restrict__ peak_off, uint32_t stride,
                                                     const float* __restrict__ sm, const float* __restrict__ si,
                                                     float* __restrict__ masses, float* __restrict__ intens) {
    const uint32_t spec = blockIdx.x, lane = threadIdx.x;
    if (spec >= n_spectra) return;
    const uint64_t p0 = peak_off[spec];
    const uint32_t c = (uint32_t)(peak_off[spec + 1] - p0);
    for (uint32_t i = lane; i < c; i += WAVE) {
        masses[p0 + i] = sm[(size_t)spec * stride + i];
        intens[p0 + i] = si[(size_t)spec * stride + i];
    }
}

}  // namespace

size_t process_lds_bytes(uint32_t rcap, uint32_t rpow2) { return ((size_t)rcap * 18 + (size_t)rpow2 * 4 + 15) & ~(size_t)15; }

int process_kernel_prepare(size_t max_lds_bytes) {
    return (int)hipFuncSetAttribute((const void*)process_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds_bytes);
}

void launch_process(uint32_t n, const uint64_t* raw_off, const float* raw_mz, const float* raw_int, const uint8_t* charge,
                    uint32_t take_top_n, bool deisotope, float min_deisotope_mz, uint32_t rcap, uint32_t rpow2, uint32_t stride,
                    float* out_mass, float* out_int, float* out_tic, uint32_t* out_count, void* stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(process_kernel<false>, dim3(n), dim3(64), process_lds_bytes(rcap, rpow2), (hipStream_t)stream, n, raw_off, raw_mz,
                       raw_int, charge, take_top_n, deisotope ? 1u : 0u, min_deisotope_mz, rcap, rpow2, stride, out_mass, out_int,
                       out_tic, out_count, (const uint32_t*)nullptr, (unsigned char*)nullptr, (size_t)0);
}

// the spectra `big_list` names (n_big of them, more raw peaks than the LDS instance takes), each in a workspace slice of
// process_lds_bytes(big_cap, big_pow2) bytes
void launch_process_big(uint32_t n_big, const uint32_t* big_list, unsigned char* workspace, const uint64_t* raw_off, const float* raw_mz,
                        const float* raw_int, const uint8_t* charge, uint32_t take_top_n, bool deisotope, float min_deisotope_mz,
                        uint32_t big_cap, uint32_t big_pow2, uint32_t stride, float* out_mass, float* out_int, float* out_tic,
                        uint32_t* out_count, void* stream) {
    if (n_big == 0) return;
    hipLaunchKernelGGL(process_kernel<true>, dim3(n_big), dim3(64), 0, (hipStream_t)stream, n_big, raw_off, raw_mz, raw_int, charge,
                       take_top_n, deisotope ? 1u : 0u, min_deisotope_mz, big_cap, big_pow2, stride, out_mass, out_int, out_tic, out_count,
                       big_list, workspace, process_lds_bytes(big_cap, big_pow2));
}

void launch_compact(uint32_t n, const uint64_t* peak_off, uint32_t stride, const float* sm, const float* si, float* masses,
                    float* intens, void* stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(compact_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, n, peak_off, stride, sm, si, masses, intens);
}

}  // namespace sagehip
