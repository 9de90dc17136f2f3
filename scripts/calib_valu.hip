// calib_valu.hip — how many cycles does a SIMD of gfx950 spend on one wave64 VALU instruction?
//
// DESIGN.md prices the two narrow-search kernels against "VALU issue": instructions per spectrum x cycles per instruction.  The
// CDNA4 guide says a wave64 f32 instruction takes two passes of a 32-wide SIMD; round 3 assumed four (SQ_ACTIVE_INST_VALU /
// SQ_INSTS_VALU ~ 1.03 quad-cycles).  This program measures it: a kernel of N independent v_add_f32 / v_fma_f32 / v_lshl_or_b32 /
// v_readlane / v_writelane / s_add_u32 / s_load_dword per wavefront (eight accumulators: no dependency stalls), and mixes of 64 v_fma_f32
// with 0..64 s_add_u32 (is the scalar pipe a co-limiter of a vector-bound kernel?  profiles/r05_valu_calibration.md), launched with W = 1..8 wavefronts per SIMD on every SIMD
// of the chip, timed with s_memtime (shader-clock cycles on gfx9) inside the kernel and with HIP events outside.
//   cycles per instruction = (cycles a SIMD was busy) / (W x N)
//
//   build: hipcc --offload-arch=gfx950 -O3 -o scripts/calib_valu scripts/calib_valu.hip     (scripts/gpu_calib_valu.sh runs it)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
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
    uint32_t s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3, s4 = s0 + 4, s5 = s0 + 5, s6 = s0 + 6, s7 = s0 + 7;
    const uint32_t sk = trips | 1u;
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
            } else if (OP == 4) {  // s_add_u32: the scalar ALU (ONE per compute unit, shared by its four SIMDs)
                asm volatile("s_add_u32 %0, %0, %8\n\ts_add_u32 %1, %1, %8\n\ts_add_u32 %2, %2, %8\n\ts_add_u32 %3, %3, %8\n\t"
                             "s_add_u32 %4, %4, %8\n\ts_add_u32 %5, %5, %8\n\ts_add_u32 %6, %6, %8\n\ts_add_u32 %7, %7, %8"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : "s"(sk) : "scc");
            } else if (OP == 5) {  // v_readlane_b32: a VALU-issued instruction with a scalar destination (what a scalar-register spill costs)
                asm volatile("v_readlane_b32 %0, %8, 1\n\tv_readlane_b32 %1, %8, 2\n\tv_readlane_b32 %2, %8, 3\n\tv_readlane_b32 %3, %8, 4\n\t"
                             "v_readlane_b32 %4, %8, 5\n\tv_readlane_b32 %5, %8, 6\n\tv_readlane_b32 %6, %8, 7\n\tv_readlane_b32 %7, %8, 8"
                             : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3), "=s"(s4), "=s"(s5), "=s"(s6), "=s"(s7) : "v"(u0));
            } else if (OP == 6) {  // v_writelane_b32
                asm volatile("v_writelane_b32 %0, %8, 1\n\tv_writelane_b32 %1, %8, 2\n\tv_writelane_b32 %2, %8, 3\n\tv_writelane_b32 %3, %8, 4\n\t"
                             "v_writelane_b32 %4, %8, 5\n\tv_writelane_b32 %5, %8, 6\n\tv_writelane_b32 %6, %8, 7\n\tv_writelane_b32 %7, %8, 8"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "s"(sk));
            } else if (OP == 7) {  // s_load_dword from the kernarg segment (an L1-scalar-cache hit), eight in flight, one wait
                asm volatile("s_load_dword %0, %8, 0x0\n\ts_load_dword %1, %8, 0x4\n\ts_load_dword %2, %8, 0x8\n\ts_load_dword %3, %8, 0xc\n\t"
                             "s_load_dword %4, %8, 0x10\n\ts_load_dword %5, %8, 0x14\n\ts_load_dword %6, %8, 0x18\n\ts_load_dword %7, %8, 0x1c\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3), "=s"(s4), "=s"(s5), "=s"(s6), "=s"(s7) : "s"(cycles) : "memory");
            } else if (OP >= 20 && OP < 40) {  // round 5, second sheet: the integer / compare / conversion instructions rescore_kernel is made of
#define U8(INSTR) asm volatile(INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7) \
                               : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(u0 ^ 5u), "v"(k) : "vcc")
#define I_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n\t"
#define I_LSHR(i) "v_lshrrev_b32 %" #i ", 1, %" #i "\n\t"
#define I_ADDU(i) "v_add_u32 %" #i ", %" #i ", %8\n\t"
#define I_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %8\n\t"
#define I_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n\t"
#define I_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %8\n\t"
#define I_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 5\n\t"
#define I_CMP(i) "v_cmp_lt_u32 vcc, %" #i ", %8\n\t"
#define I_CMPSEL(i) "v_cmp_lt_u32 vcc, %" #i ", %8\n\tv_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\t"
#define I_CVT(i) "v_cvt_u32_f32 %" #i ", %" #i "\n\t"
#define I_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n\t"
#define I_FFBL(i) "v_ffbl_b32 %" #i ", %" #i "\n\t"
#define I_BCNT(i) "v_bcnt_u32_b32 %" #i ", %" #i ", %8\n\t"
#define I_MBCNT(i) "v_mbcnt_lo_u32_b32 %" #i ", %" #i ", %8\n\t"
#define I_MULF(i) "v_mul_f32 %" #i ", %" #i ", %9\n\t"
#define I_MAXF(i) "v_max_f32 %" #i ", %" #i ", %9\n\t"
#define I_ADDCO(i) "v_add_co_u32 %" #i ", vcc, %" #i ", %8\n\t"
#define I_XOR3(i) "v_xad_u32 %" #i ", %" #i ", %8, %8\n\t"
#define I_DSREAD(i) "ds_read_b32 %" #i ", %8\n\t"
                if (OP == 20) U8(I_AND);
                else if (OP == 21) U8(I_LSHR);
                else if (OP == 22) U8(I_ADDU);
                else if (OP == 23) U8(I_MULHI);
                else if (OP == 24) U8(I_MULLO);
                else if (OP == 25) U8(I_MAD24);
                else if (OP == 26) U8(I_BFE);
                else if (OP == 27) U8(I_CMP);
                else if (OP == 28) U8(I_CMPSEL);  // (two instructions per item: the row counts 128 per group)
                else if (OP == 29) U8(I_CVT);
                else if (OP == 30) U8(I_RCP);
                else if (OP == 31) U8(I_FFBL);
                else if (OP == 32) U8(I_BCNT);
                else if (OP == 33) U8(I_MBCNT);
                else if (OP == 34) U8(I_MULF);
                else if (OP == 35) U8(I_MAXF);
                else if (OP == 36) U8(I_ADDCO);
                else if (OP == 37) U8(I_XOR3);
                else if (OP == 38) {  // LDS reads issued by the vector memory path: eight in flight, one wait
                    asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:256\n\tds_read_b32 %2, %8 offset:512\n\tds_read_b32 %3, %8 offset:768\n\t"
                                 "ds_read_b32 %4, %8 offset:1024\n\tds_read_b32 %5, %8 offset:1280\n\tds_read_b32 %6, %8 offset:1536\n\tds_read_b32 %7, %8 offset:1792\n\t"
                                 "s_waitcnt lgkmcnt(0)"
                                 : "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3), "=v"(u4), "=v"(u5), "=v"(u6), "=v"(u7) : "v"((uint32_t)threadIdx.x * 4u) : "memory");
                } else {  // 39: 64-bit shift (two registers per operand: four accumulators, twice)
                    uint64_t q0 = u0, q1 = u1, q2 = u2, q3 = u3;
                    asm volatile("v_lshlrev_b64 %0, 1, %0\n\tv_lshlrev_b64 %1, 1, %1\n\tv_lshlrev_b64 %2, 1, %2\n\tv_lshlrev_b64 %3, 1, %3\n\t"
                                 "v_lshlrev_b64 %0, 1, %0\n\tv_lshlrev_b64 %1, 1, %1\n\tv_lshlrev_b64 %2, 1, %2\n\tv_lshlrev_b64 %3, 1, %3"
                                 : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
                    u0 = (uint32_t)q0 | 1u; u1 = (uint32_t)q1 | 1u; u2 = (uint32_t)q2 | 1u; u3 = (uint32_t)q3 | 1u;
                }
            } else if (OP >= 10) {  // the rescoring kernel's MIX: eight v_fma_f32 + (OP - 10) s_add_u32 per group (64 : 0 / 16 / 32 / 48 / 64)
                asm volatile("v_fma_f32 %0, %0, %8, %8\n\tv_fma_f32 %1, %1, %8, %8\n\tv_fma_f32 %2, %2, %8, %8\n\tv_fma_f32 %3, %3, %8, %8\n\t"
                             "v_fma_f32 %4, %4, %8, %8\n\tv_fma_f32 %5, %5, %8, %8\n\tv_fma_f32 %6, %6, %8, %8\n\tv_fma_f32 %7, %7, %8, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
                if (OP - 10 >= 2) asm volatile("s_add_u32 %0, %0, %2\n\ts_add_u32 %1, %1, %2" : "+s"(s0), "+s"(s1) : "s"(sk) : "scc");
                if (OP - 10 >= 4) asm volatile("s_add_u32 %0, %0, %2\n\ts_add_u32 %1, %1, %2" : "+s"(s2), "+s"(s3) : "s"(sk) : "scc");
                if (OP - 10 >= 6) asm volatile("s_add_u32 %0, %0, %2\n\ts_add_u32 %1, %1, %2" : "+s"(s4), "+s"(s5) : "s"(sk) : "scc");
                if (OP - 10 >= 8) asm volatile("s_add_u32 %0, %0, %2\n\ts_add_u32 %1, %1, %2" : "+s"(s6), "+s"(s7) : "s"(sk) : "scc");
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
    const float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7) + (float)(s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7);
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
    struct Op { const char* name; int id; double vector_instr, scalar_instr; };
    const Op ops[] = {{"v_add_f32", 0, 64, 0}, {"v_fma_f32", 1, 64, 0}, {"v_lshl_or_b32", 2, 64, 0}, {"v_fma_f64", 3, 64, 0},
                      {"s_add_u32", 4, 0, 64}, {"v_readlane_b32", 5, 64, 0}, {"v_writelane_b32", 6, 64, 0}, {"s_load_dword (8 in flight + wait)", 7, 0, 64},
                      {"mix 64 v_fma_f32 + 0 s_add_u32", 10, 64, 0}, {"mix 64 v_fma_f32 + 16 s_add_u32", 12, 64, 16},
                      {"mix 64 v_fma_f32 + 32 s_add_u32", 14, 64, 32}, {"mix 64 v_fma_f32 + 48 s_add_u32", 16, 64, 48},
                      {"mix 64 v_fma_f32 + 64 s_add_u32", 18, 64, 64},
                      {"v_and_b32", 20, 64, 0}, {"v_lshrrev_b32", 21, 64, 0}, {"v_add_u32", 22, 64, 0}, {"v_mul_hi_u32", 23, 64, 0},
                      {"v_mul_lo_u32", 24, 64, 0}, {"v_mad_u32_u24", 25, 64, 0}, {"v_bfe_u32", 26, 64, 0}, {"v_cmp_lt_u32", 27, 64, 0},
                      {"v_cmp_lt_u32 + v_cndmask_b32", 28, 128, 0}, {"v_cvt_u32_f32", 29, 64, 0}, {"v_rcp_f32", 30, 64, 0}, {"v_ffbl_b32", 31, 64, 0},
                      {"v_bcnt_u32_b32", 32, 64, 0}, {"v_mbcnt_lo_u32_b32", 33, 64, 0}, {"v_mul_f32", 34, 64, 0}, {"v_max_f32", 35, 64, 0},
                      {"v_add_co_u32", 36, 64, 0}, {"v_xad_u32", 37, 64, 0}, {"ds_read_b32 (8 in flight + wait)", 38, 64, 0},
                      {"v_lshlrev_b64", 39, 64, 0}};
    const bool second_sheet_only = std::getenv("CALIB_SECOND_SHEET") != nullptr;
    std::printf("| instruction | waves/SIMD | instr per wave (vector + scalar) | wave cycles (s_memtime, median) | SIMD cycles per 64-instruction "
                "group per wave (s_memtime) | wall ms | CU cycles per group at clockRate (wall), all four SIMDs busy |\n|---|---|---|---|---|---|---|\n");
    for (const Op& op : ops)
        for (int w : {1, 2, 4, 5, 8}) {
            if (second_sheet_only && (op.id < 20 || (w != 1 && w != 5))) continue;
            const int blocks = n_simd * w;  // single-wavefront workgroups: the dispatcher deals them round-robin over CUs and SIMDs
            auto launch = [&]() {
                switch (op.id) {
#define CASE(N) case N: hipLaunchKernelGGL(valu_kernel<N>, dim3(blocks), dim3(64), 0, 0, trips, sink, cyc); break;
                    CASE(0) CASE(1) CASE(2) CASE(4) CASE(5) CASE(6) CASE(7) CASE(10) CASE(12) CASE(14) CASE(16) CASE(18)
                    CASE(20) CASE(21) CASE(22) CASE(23) CASE(24) CASE(25) CASE(26) CASE(27) CASE(28) CASE(29) CASE(30) CASE(31) CASE(32)
                    CASE(33) CASE(34) CASE(35) CASE(36) CASE(37) CASE(38) CASE(39)
#undef CASE
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
            const double groups = (double)trips * UNROLL / 64.0;  // one group = 64 vector and / or the row's scalar instructions
            const double wave_cycles = (double)h[blocks / 2];
            std::printf("| %s | %d | %.0f + %.0f | %.0f | %.1f | %.3f | %.1f |\n", op.name, w, groups * op.vector_instr, groups * op.scalar_instr,
                        wave_cycles, wave_cycles / (groups * w), ms, ms * 1e-3 * prop.clockRate * 1e3 / (groups * w));
        }
    return 0;
}
