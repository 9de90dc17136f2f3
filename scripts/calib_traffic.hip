// calib_traffic.hip — known-byte access patterns for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950.
//
// MI355X_MICROARCH.md §HBM calibrates FETCH_SIZE only for 16 B/lane coalesced streams (it reads 1/2 of the bytes there) and
// says other widths are uncalibrated.  The search kernels mostly do 4 B table gathers and short 16 B-per-lane runs, so this
// program issues those patterns over a 4 GiB array (16x the 256 MiB Infinity Cache) with a known number of requested bytes
// and touched sectors; run it under `rocprofv3 --pmc FETCH_SIZE` (and WRITE_SIZE, TCC_EA0_RDREQ..., in separate passes) and
// feed the per-kernel counters + the lines this program prints to scripts/calib_summary.py.
//
//   build: hipcc --offload-arch=gfx950 -O3 -o scripts/calib_traffic scripts/calib_traffic.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            std::exit(1);                                                      \
        }                                                                      \
    } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {  // splitmix64: a fixed pseudo-random permutation-like hash
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// 1. coalesced stream, 16 B per lane (the pattern the guide calibrated)
__global__ void stream_read16(const uint4* __restrict__ a, uint64_t n16, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 v = a[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// 2. coalesced stream, 4 B per lane
__global__ void stream_read4(const uint32_t* __restrict__ a, uint64_t n4, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * blockDim.x) acc += a[i];
    if (acc == 0x12345678u) sink[0] = acc;
}
// 3. random gather of W-byte words (W = 4, 8, 16): one access per lane per trip, every lane an independent random address
template <typename T>
__global__ void gather_read(const T* __restrict__ a, uint64_t nwords, uint64_t accesses, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < accesses; i += (uint64_t)gridDim.x * blockDim.x) {
        const T v = a[mix(i) % nwords];
        acc += ((const uint32_t*)&v)[0];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// 4. the search kernels' entry reads: an 8-lane group reads RUN consecutive 16 B cells starting at a random 16 B-aligned
//    position (RUN = 8: one 128 B run per group per trip)
__global__ void gather_runs(const uint4* __restrict__ a, uint64_t n16, uint64_t groups, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t sub = (uint32_t)(tid & 7u);
    for (uint64_t g = tid >> 3; g < groups; g += ((uint64_t)gridDim.x * blockDim.x) >> 3) {
        const uint64_t base = mix(g) % (n16 - 8);
        const uint4 v = a[base + sub];
        acc += v.x ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// 5. writes: coalesced 16 B per lane, and random 4 B scatter
__global__ void stream_write16(uint4* __restrict__ a, uint64_t n16) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x)
        a[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
__global__ void scatter_write4(uint32_t* __restrict__ a, uint64_t n4, uint64_t accesses) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < accesses; i += (uint64_t)gridDim.x * blockDim.x)
        a[mix(i) % n4] = (uint32_t)i;
}

int main(int argc, char** argv) {
    const uint64_t bytes = (argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 4096ull) << 20;  // MiB, default 4 GiB
    const uint64_t accesses = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : (16ull << 20);  // random accesses per gather kernel
    void* a = nullptr;
    uint32_t* sink = nullptr;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc((void**)&sink, 64));
    CK(hipMemset(a, 1, bytes));
    CK(hipDeviceSynchronize());
    const dim3 grid(256 * 8), block(256);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timed = [&](const char* name, double req_bytes, double sectors64, auto launch) {
        launch();  // warm-up (TLB; the array itself is far larger than every cache)
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("CALIB %s requested_bytes=%.0f sectors64_bytes=%.0f ms=%.4f\n", name, req_bytes, sectors64 * 64.0, ms);
    };
    const uint64_t n16 = bytes / 16, n4 = bytes / 4;
    timed("stream_read16", (double)bytes, (double)bytes / 64, [&] { hipLaunchKernelGGL(stream_read16, grid, block, 0, 0, (const uint4*)a, n16, sink); });
    timed("stream_read4", (double)bytes, (double)bytes / 64, [&] { hipLaunchKernelGGL(stream_read4, grid, block, 0, 0, (const uint32_t*)a, n4, sink); });
    // (random accesses: `accesses` distinct-with-high-probability sectors; collisions inside L2's reach are negligible at 16 M over 4 GiB)
    timed("gather_read4", 4.0 * accesses, (double)accesses, [&] { hipLaunchKernelGGL(gather_read<uint32_t>, grid, block, 0, 0, (const uint32_t*)a, n4, accesses, sink); });
    timed("gather_read8", 8.0 * accesses, (double)accesses, [&] { hipLaunchKernelGGL(gather_read<uint2>, grid, block, 0, 0, (const uint2*)a, bytes / 8, accesses, sink); });
    timed("gather_read16", 16.0 * accesses, (double)accesses, [&] { hipLaunchKernelGGL(gather_read<uint4>, grid, block, 0, 0, (const uint4*)a, n16, accesses, sink); });
    // a 128 B run at a random 16 B-aligned offset straddles 64 B sectors: 2 + 15/4 ... on average (128 + 48) / 64 = 2.75 sectors
    timed("gather_runs128", 128.0 * (accesses / 8), 2.75 * (accesses / 8), [&] { hipLaunchKernelGGL(gather_runs, grid, block, 0, 0, (const uint4*)a, n16, accesses / 8, sink); });
    timed("stream_write16", (double)bytes, (double)bytes / 64, [&] { hipLaunchKernelGGL(stream_write16, grid, block, 0, 0, (uint4*)a, n16); });
    timed("scatter_write4", 4.0 * accesses, (double)accesses, [&] { hipLaunchKernelGGL(scatter_write4, grid, block, 0, 0, (uint32_t*)a, n4, accesses); });
    CK(hipFree(a));
    CK(hipFree(sink));
    return 0;
}
