// calib_valu.hip — how many cycles does a SIMD of gfx950 spend on one wave64 VALU instruction?
//
// DESIGN.md prices the two narrow-search kernels against "VALU issue": instructions per spectrum x cycles per instruction.  The
// CDNA4 guide says a wave64 f32 instruction takes two passes of a 32-wide SIMD; round 3 assumed four (SQ_ACTIVE_INST_VALU /
// SQ_INSTS_VALU ~ 1.03 quad-cycles).  This program measures it: a kernel of N independent v_add_f32 / v_fma_f32 / v_lshl_or_b32 /
// v_readlane per wavefront (eight accumulators: no dependency stalls), launched with W = 1..8 wavefronts per SIMD on every SIMD
// of the chip, timed with s_memtime (shader-clock cycles on gfx9) inside the kernel and with HIP events outside.
//   cycles per instruction = (cycles a SIMD was busy) / (W x N)
//
//   build: hipcc --offload-arch=gfx950 -O3 -o scripts/calib_valu scripts/calib_valu.hip     (scripts/gpu_calib_valu.sh runs it)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                       \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) {                                                                     \
            std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);     \
            std::exit(1);                                                                           \
        }                                                                                           \
    } while (0)

constexpr int UNROLL = 64;  // instructions per loop trip (8 accumulators x 8)

template <int OP>
__global__ __launch_bounds__(64) void valu_kernel(uint32_t trips, float* sink, unsigned long long* cycles) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
    const float k = 1.0000001f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t t = 0; t < trips; t++) {
#pragma unroll
        for (int i = 0; i < UNROLL / 8; i++) {
            if (OP == 0) {  // v_add_f32
                asm volatile("v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %8\n\tv_add_f32 %2, %2, %8\n\tv_add_f32 %3, %3, %8\n\t"
                             "v_add_f32 %4, %4, %8\n\tv_add_f32 %5, %5, %8\n\tv_add_f32 %6, %6, %8\n\tv_add_f32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
            } else if (OP == 1) {  // v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %8, %8\n\tv_fma_f32 %1, %1, %8, %8\n\tv_fma_f32 %2, %2, %8, %8\n\tv_fma_f32 %3, %3, %8, %8\n\t"
                             "v_fma_f32 %4, %4, %8, %8\n\tv_fma_f32 %5, %5, %8, %8\n\tv_fma_f32 %6, %6, %8, %8\n\tv_fma_f32 %7, %7, %8, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
            } else if (OP == 2) {  // v_lshl_or_b32 (integer, VOP3)
                asm volatile("v_lshl_or_b32 %0, %0, 1, %8\n\tv_lshl_or_b32 %1, %1, 1, %8\n\tv_lshl_or_b32 %2, %2, 1, %8\n\tv_lshl_or_b32 %3, %3, 1, %8\n\t"
                             "v_lshl_or_b32 %4, %4, 1, %8\n\tv_lshl_or_b32 %5, %5, 1, %8\n\tv_lshl_or_b32 %6, %6, 1, %8\n\tv_lshl_or_b32 %7, %7, 1, %8"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(u0 ^ 5u));
            } else {  // v_fma_f64 (two registers per operand: four accumulators, counted as eight instructions of half the unroll)
                double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
                const double kd = 1.0000001;
                asm volatile("v_fma_f64 %0, %0, %4, %4\n\tv_fma_f64 %1, %1, %4, %4\n\tv_fma_f64 %2, %2, %4, %4\n\tv_fma_f64 %3, %3, %4, %4\n\t"
                             "v_fma_f64 %0, %0, %4, %4\n\tv_fma_f64 %1, %1, %4, %4\n\tv_fma_f64 %2, %2, %4, %4\n\tv_fma_f64 %3, %3, %4, %4"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(kd));
                a0 = (float)d0; a1 = (float)d1; a2 = (float)d2; a3 = (float)d3;
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7);
    if (s == 1.2345e-30f) sink[0] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_simd = prop.multiProcessorCount * 4;
    std::printf("# %s: %d CUs, %d SIMDs, clockRate %.0f MHz\n", prop.name, prop.multiProcessorCount, n_simd, prop.clockRate / 1e3);
    float* sink;
    unsigned long long* cyc;
    CK(hipMalloc(&sink, 4));
    CK(hipMalloc(&cyc, (size_t)n_simd * 8 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const uint32_t trips = 20000;
    const char* names[4] = {"v_add_f32", "v_fma_f32", "v_lshl_or_b32", "v_fma_f64"};
    std::printf("| instruction | waves/SIMD | instr per wave | wave cycles (s_memtime, median) | cycles per instr per SIMD (s_memtime) | wall ms | "
                "cycles per instr at clockRate (wall) |\n|---|---|---|---|---|---|---|\n");
    for (int op = 0; op < 4; op++)
        for (int w = 1; w <= 8; w *= 2) {
            const int blocks = n_simd * w;  // single-wavefront workgroups: the dispatcher deals them round-robin over CUs and SIMDs
            auto launch = [&]() {
                switch (op) {
                    case 0: hipLaunchKernelGGL(valu_kernel<0>, dim3(blocks), dim3(64), 0, 0, trips, sink, cyc); break;
                    case 1: hipLaunchKernelGGL(valu_kernel<1>, dim3(blocks), dim3(64), 0, 0, trips, sink, cyc); break;
                    case 2: hipLaunchKernelGGL(valu_kernel<2>, dim3(blocks), dim3(64), 0, 0, trips, sink, cyc); break;
                    default: hipLaunchKernelGGL(valu_kernel<3>, dim3(blocks), dim3(64), 0, 0, trips, sink, cyc); break;
                }
            };
            launch();  // warm-up (clocks)
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            launch();
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> h(blocks);
            CK(hipMemcpy(h.data(), cyc, (size_t)blocks * 8, hipMemcpyDeviceToHost));
            std::nth_element(h.begin(), h.begin() + blocks / 2, h.end());
            const double n_instr = (double)trips * UNROLL;
            const double wave_cycles = (double)h[blocks / 2];
            std::printf("| %s | %d | %.0f | %.0f | %.2f | %.3f | %.2f |\n", names[op], w, n_instr, wave_cycles, wave_cycles / (n_instr * w), ms,
                        ms * 1e-3 * prop.clockRate * 1e3 / (n_instr * w));
        }
    return 0;
}
