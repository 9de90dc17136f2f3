// detmath_oracle.h — a COPY of sage_amd/csrc/detmath.h (the arithmetic contract of the post-search rescoring) with the namespace
// and include guard renamed, so that the checker does not include product headers; oracle/selftest.cpp holds the two to each
// other bit for bit and to the platform libm within 1 ulp.  It is not an independent implementation: comparing the device with
// the oracle's `det` mode proves that the device evaluates the stated contract; the independent check of the contract itself is
// the oracle's other mode (platform libm, every sum sequential — the reference's own arithmetic), against which the device's
// coefficients, discriminants, posterior errors and q-values are compared in tests/test_gpu_rescore.py.
// Test infrastructure: nothing under sage_amd/ includes this file.
//
// Why this exists.  Sage's LDA fit (crates/sage/src/ml/linear_discriminant.rs:57-127) ends in a Gauss-Jordan elimination whose
// pivot search compares matrix entries with `>=` and `== 0.0` (ml/gauss.rs:89-124) and whose success test is exact
// (`x != 1.0 && x != 0.0`, gauss.rs:69-87).  With the constant columns every non-ion-mobility search produces (ims == 0,
// delta_ims_model == 0.999) some of those entries are pure rounding noise, so whether the model is fitted or the heuristic
// discriminant of runner.rs:285-288 is used can hinge on the last bit of a sum.  Two evaluations of that pipeline agree on the
// branch — and then on every discriminant, q-value and posterior error — only if they agree on every bit that enters the
// elimination.  The reference itself does not define those bits: Kde::pdf sums in rayon's work-stealing order (ml/kde.rs:38-46)
// and ln_1p / exp are whatever the platform libm returns.  So the product FIXES them, here:
//
//   * det_log1p / det_exp / det_log1pf: ln_1p and exp from IEEE +, -, *, / and integer operations on the bit pattern only
//     (the classic argument-reduction + minimax-polynomial algorithms, < 1 ulp).  No libm call, no FMA (the translation
//     units are compiled with -ffp-contract=off), so gfx950 and x86-64 return the same bits for the same input.
//   * the BLOCKED ORDER of the long f64 reductions whose order the reference itself leaves open or that do not feed the
//     elimination: the kernel-density sums (Kde::pdf: rayon's order) and the bandwidth statistics.  Elements are cut into
//     consecutive blocks of DET_BLOCK; inside a block the sum runs left to right from +0.0; the block sums are then added left
//     to right from +0.0.  For n <= DET_BLOCK this IS the sequential `iter().sum()`.
//   * NOT the LDA sums (class means, within-class scatter, linear_discriminant.rs:70-103): those run strictly in row order, one
//     running sum per accumulator, exactly as the reference's loops do (rescore.hip: seq_lda_kernel).
//
// The CPU checker (oracle/rescore_oracle.cpp) keeps its own copy of these functions (oracle/detmath_oracle.h — the checker does
// not include product sources) and evaluates either the contract (`det` mode) or the reference's own order with the platform
// libm; tests hold the device bit-exactly to the former, compare it with the latter (coefficients to 1e-9, same fit/fall-back
// decision, q-values), and oracle/selftest.cpp measures both copies against the platform libm and against each other.
//
// det_log1p and det_exp follow the algorithms of FreeBSD msun / fdlibm's s_log1p.c and e_exp.c (constants, thresholds and
// evaluation order), restated for both x86-64 and gfx950.  Their notice, as that licence requires:
//   ====================================================
//   Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
//
//   Developed at SunPro, a Sun Microsystems, Inc. business.
//   Permission to use, copy, modify, and distribute this
//   software is freely granted, provided that this notice
//   is preserved.
//   ====================================================
#pragma once
#include <stdint.h>

#define ORC_DM inline

namespace orcdet {

constexpr uint32_t DET_BLOCK = 1024;  // elements per block of the blocked summation order

ORC_DM uint64_t d2u(double d) {
    union { double d; uint64_t u; } c;
    c.d = d;
    return c.u;
}
ORC_DM double u2d(uint64_t u) {
    union { double d; uint64_t u; } c;
    c.u = u;
    return c.d;
}
ORC_DM int32_t hi_word(double d) { return (int32_t)(d2u(d) >> 32); }
ORC_DM double with_hi_word(double d, int32_t hi) { return u2d((d2u(d) & 0xFFFFFFFFull) | ((uint64_t)(uint32_t)hi << 32)); }

// ln(1 + x).  Argument reduction 1 + x = 2^k (1 + f), sqrt(2)/2 < 1 + f < sqrt(2), with the rounding error of 1 + x
// carried as a correction term; log(1 + f) = f - f^2/2 + s (f^2/2 + R(s^2)), s = f / (2 + f), R a degree-7 minimax polynomial.
ORC_DM double det_log1p(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10, two54 = 1.80143985094819840000e+16;
    const double Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01, Lp3 = 2.857142874366239149e-01,
                 Lp4 = 2.222219843214978396e-01, Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01,
                 Lp7 = 1.479819860511658591e-01;
    double hfsq, f = 0.0, c = 0.0, s, z, R, u;
    int32_t k = 1, hu = 0;
    const int32_t hx = hi_word(x), ax = hx & 0x7fffffff;
    if (hx < 0x3FDA827A) {          // x < 0.41422
        if (ax >= 0x3ff00000) {     // x <= -1.0
            const double zero = 0.0;
            if (x == -1.0) return -two54 / zero;  // -inf
            return (x - x) / (x - x);             // NaN
        }
        if (ax < 0x3e200000) {      // |x| < 2^-29
            if (ax < 0x3c900000) return x;  // |x| < 2^-54
            return x - x * x * 0.5;
        }
        if (hx > 0 || hx <= (int32_t)0xbfd2bec3) {  // -0.2929 < x < 0.41422
            k = 0;
            f = x;
            hu = 1;
        }
    }
    if (hx >= 0x7ff00000) return x + x;  // +inf, NaN
    if (k != 0) {
        if (hx < 0x43400000) {
            u = 1.0 + x;
            hu = hi_word(u);
            k = (hu >> 20) - 1023;
            c = (k > 0) ? 1.0 - (u - x) : x - (u - 1.0);  // the rounding error of 1 + x
            c /= u;
        } else {
            u = x;
            hu = hi_word(u);
            k = (hu >> 20) - 1023;
            c = 0.0;
        }
        hu &= 0x000fffff;
        if (hu < 0x6a09e) {
            u = with_hi_word(u, hu | 0x3ff00000);  // normalise u
        } else {
            k += 1;
            u = with_hi_word(u, hu | 0x3fe00000);  // normalise u / 2
            hu = (0x00100000 - hu) >> 2;
        }
        f = u - 1.0;
    }
    hfsq = 0.5 * f * f;
    const double dk = (double)k;
    if (hu == 0) {  // |f| < 2^-20
        if (f == 0.0) {
            if (k == 0) return 0.0;
            c += dk * ln2_lo;
            return dk * ln2_hi + c;
        }
        R = hfsq * (1.0 - 0.66666666666666666 * f);
        if (k == 0) return f - R;
        return dk * ln2_hi - ((R - (dk * ln2_lo + c)) - f);
    }
    s = f / (2.0 + f);
    z = s * s;
    R = z * (Lp1 + z * (Lp2 + z * (Lp3 + z * (Lp4 + z * (Lp5 + z * (Lp6 + z * Lp7))))));
    if (k == 0) return f - (hfsq - s * (hfsq + R));
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + (dk * ln2_lo + c))) - f);
}

// f32 ln_1p (runner.rs:287 `(-poisson as f32).ln_1p()`): evaluated in f64 and rounded once
ORC_DM float det_log1pf(float x) { return (float)det_log1p((double)x); }

// e^x.  x = k ln2 + r, |r| <= 0.5 ln2 (ln2 split in two so that k ln2_hi is exact); e^r = 1 + 2r / (R(r^2) - r) with a
// degree-5 minimax polynomial; scaled by 2^k through the exponent field.
ORC_DM double det_exp(double x) {
    const double o_threshold = 7.09782712893383973096e+02, u_threshold = -7.45133219101941108420e+02;
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10, invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    const double huge = 1.0e+300, twom1000 = 9.33263618503218878990e-302;  // 2^-1000
    double hi = 0.0, lo = 0.0, c, t, y;
    int32_t k = 0;
    const uint64_t bits = d2u(x);
    const int32_t xsb = (int32_t)(bits >> 63);
    const uint32_t hx = (uint32_t)(bits >> 32) & 0x7fffffffu;
    if (hx >= 0x40862E42u) {  // |x| >= 709.78
        if (hx >= 0x7ff00000u) {
            if (((hx & 0xfffffu) | (uint32_t)bits) != 0) return x + x;  // NaN
            return xsb == 0 ? x : 0.0;                                  // exp(+-inf)
        }
        if (x > o_threshold) return huge * huge;          // overflow -> +inf
        if (x < u_threshold) return twom1000 * twom1000;  // underflow -> 0
    }
    if (hx > 0x3fd62e42u) {       // |x| > 0.5 ln2
        if (hx < 0x3FF0A2B2u) {   // and |x| < 1.5 ln2
            hi = xsb ? x + ln2_hi : x - ln2_hi;
            lo = xsb ? -ln2_lo : ln2_lo;
            k = 1 - xsb - xsb;
        } else {
            k = (int32_t)(invln2 * x + (xsb ? -0.5 : 0.5));
            t = (double)k;
            hi = x - t * ln2_hi;  // t * ln2_hi is exact
            lo = t * ln2_lo;
        }
        x = hi - lo;
    } else if (hx < 0x3e300000u) {  // |x| < 2^-28
        return 1.0 + x;
    }
    t = x * x;
    c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
    y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
    if (k >= -1021) return u2d(d2u(y) + ((uint64_t)(uint32_t)k << 52));  // y * 2^k through the exponent field
    return u2d(d2u(y) + ((uint64_t)(uint32_t)(k + 1000) << 52)) * twom1000;
}

// Blocked-order sum of term(i), i in [0, n): the reference semantics of `iter().sum()` / fold(0.0, +) with the order the
// device evaluates (see the header).  Host-side statement of the contract (the device has one chain per block).
template <class Term>
inline double blocked_sum(uint64_t n, Term term) {
    double total = 0.0;
    for (uint64_t b0 = 0; b0 < n; b0 += DET_BLOCK) {
        const uint64_t b1 = b0 + DET_BLOCK < n ? b0 + DET_BLOCK : n;
        double part = 0.0;
        for (uint64_t i = b0; i < b1; ++i) part += term(i);
        total += part;
    }
    return total;
}

}  // namespace orcdet
