// The exhaustive proof behind core.h's div_const_fast: for c = 1e6 and c = 3, over ALL 2^32 f32 dividends x, compare
//     q0 = x * RN(1 / c);  e = fma(-q0, c, x);  q = fma(e, RN(1 / c), q0)
// with the IEEE quotient x / c, bit for bit, and print what differs: the count, how many of them differ in the sign of a zero
// only, whether +-inf are among them, and the largest finite |x| that differs.  tests/test_core_emulation.py builds and runs it
// and holds the output against what core.h claims (FAST_DIV_LO .. FAST_DIV_HI free of differences).  TEST INFRASTRUCTURE.
//   gcc -O2 -mfma -ffp-contract=off -fopenmp div_const_proof.c -o div_const_proof -lm
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define BLOCK 4096

static void prove(const float c) {
    const float rc = 1.0f / c;
    uint64_t differ = 0, zero_sign = 0, infinite = 0;
    uint32_t largest_finite = 0;
#pragma omp parallel for schedule(static) reduction(+ : differ, zero_sign, infinite) reduction(max : largest_finite)
    for (int64_t b = 0; b < (1ll << 32) / BLOCK; b++) {
        float x[BLOCK], want[BLOCK], got[BLOCK];
        uint32_t u[BLOCK];
        for (int i = 0; i < BLOCK; i++) u[i] = (uint32_t)(b * BLOCK + i);
        memcpy(x, u, sizeof(x));
        for (int i = 0; i < BLOCK; i++) want[i] = x[i] / c;
        for (int i = 0; i < BLOCK; i++) {
            const float q0 = x[i] * rc;
            const float e = fmaf(-q0, c, x[i]);
            got[i] = fmaf(e, rc, q0);
        }
        uint32_t w[BLOCK], g[BLOCK];
        memcpy(w, want, sizeof(w));
        memcpy(g, got, sizeof(g));
        int any = 0;
        for (int i = 0; i < BLOCK; i++) any |= w[i] != g[i];
        if (!any) continue;
        for (int i = 0; i < BLOCK; i++) {
            if (w[i] == g[i] || (want[i] != want[i] && got[i] != got[i])) continue;  // (a NaN for a NaN)
            differ++;
            const uint32_t a = u[i] & 0x7FFFFFFFu;
            if (want[i] == 0.0f && got[i] == 0.0f) zero_sign++;
            else if (a == 0x7F800000u) infinite++;
            else if (a > largest_finite) largest_finite = a;
        }
    }
    float lf;
    memcpy(&lf, &largest_finite, 4);
    printf("c=%g differ=%llu zero_sign_only=%llu infinite=%llu largest_finite_abs_x=%a bits=0x%08x\n", (double)c, (unsigned long long)differ,
           (unsigned long long)zero_sign, (unsigned long long)infinite, (double)lf, largest_finite);
}

int main(void) {
    prove(1000000.0f);
    prove(3.0f);
    return 0;
}
