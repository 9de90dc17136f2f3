// crlog.h — the natural logarithm of a double, correctly rounded (round to nearest even), shared by the device kernels and the
// host (SAGE_HD): the `ln()` of the hyperscore (scoring.rs:179-201), of lnfact's Stirling series (scoring.rs:170-177) and of
// the Poisson term (scoring.rs:522-523).
//
// Why: the reference calls the platform's libm.  A PSM's hyperscore is `ln(i) + lnfact(b) + lnfact(y)`, and PSMs are RANKED by
// it — two libms that differ in the last bit of ln(i) for one candidate can order two near-equal sums differently (DESIGN.md
// 4.8).  glibc >= 2.28 rounds ln correctly for 99.99 % of the arguments this path produces (measured, tests/test_crlog.py);
// glibc < 2.28 (IBM accurate mathematical library), CORE-MATH and every other correctly rounding libm do for all of them.  The
// device's ocml `log` differs from either in ~10 % of the arguments.  A correctly rounded result is the one contract that
// does not depend on the platform, so that is what the product computes, on the device AND on the host (the lnfact table).
//
// Method (Ziv's two-phase strategy):
//   x = 2^e m, m in [1, 2); cell k = top 7 mantissa bits; c_k a multiple of 2^-8 near 1/m (crlog_tables.h), so that
//   r = fma(m, c_k, -1) is EXACT (m c_k is a multiple of 2^-60 and |r| <= 2^-7: 53 bits) and
//       ln x = e' ln2 + lc_k + log1p(r),     e' = e + (k >= SPLIT),  lc_k = -ln(c_k) resp. -ln(2 c_k)  (double-double).
//   The three terms cancel nowhere: e' != 0 gives |ln x| >= 0.34; e' == 0 with lc_k != 0 gives |ln x| >= 2^-7.01; in the two
//   cells next to 1 (k = 0 with e = 0, k = 127 with e = -1) both leading terms are exactly zero and ln x = log1p(r), r = x - 1.
//   fast phase: exact leading sum by TwoSum with r^2, r^3 and r^4 as exact products, the series from r^5 on in double; relative
//               error < 2^-81 (bound used: 2^-79).  If both ends of the error interval round to the same double, that is the
//               result — all but ~2^-25 of the arguments.  The hot rescoring kernel carries ONLY this phase and sends a
//               spectrum with an undecided logarithm through the retry pass, whose kernels carry both (kernels.hip).
//   accurate phase: the series up to r^8 in double-double, 9..18 in double, ln2 as a triple; relative error < 2^-98, so the
//               rounding is wrong only when ln x lies within 2^-98 |ln x| of a midpoint of two doubles (probability ~2^-44 per
//               call; no such argument is known and the hardest-to-round cases of ln need ~2^-118 — this is "correctly
//               rounded" in the sense of every practical libm that claims it, not a proof for all 2^63 arguments).
// tests/test_crlog.py holds both phases (each forced) to a 90-digit decimal reference on millions of arguments of this path.
#pragma once
#include <stdint.h>
#include <string.h>

#include "crlog_tables.h"

#ifndef SAGE_HD
#if defined(__HIPCC__)
#define SAGE_HD __host__ __device__ __forceinline__
#else
#define SAGE_HD inline
#endif
#endif

namespace sagecore {

struct CrLogCell {
    double c, lh, ll;
};
#if defined(__HIP_DEVICE_COMPILE__)
__device__ const CrLogCell CRLOG_TABLE[128] = {SAGE_CRLOG_TABLE};
#else
static const CrLogCell CRLOG_TABLE[128] = {SAGE_CRLOG_TABLE};
#endif

struct dd {
    double h, l;
};
SAGE_HD double crl_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
SAGE_HD dd two_sum(double a, double b) {  // a + b exactly
    const double s = a + b, bb = s - a;
    return dd{s, (a - (s - bb)) + (b - bb)};
}
SAGE_HD dd fast_two_sum(double a, double b) {  // |a| >= |b| (or a == 0)
    const double s = a + b;
    return dd{s, b - (s - a)};
}
SAGE_HD dd two_prod(double a, double b) {
    const double p = a * b;
    return dd{p, crl_fma(a, b, -p)};
}
SAGE_HD dd dd_add(dd a, dd b) {  // relative error <= 2^-104 (Dekker / Knuth "accurate" sum)
    dd s = two_sum(a.h, b.h);
    const dd t = two_sum(a.l, b.l);
    s.l += t.h;
    s = fast_two_sum(s.h, s.l);
    s.l += t.l;
    return fast_two_sum(s.h, s.l);
}
SAGE_HD dd dd_add_d(dd a, double b) {
    dd s = two_sum(a.h, b);
    s.l += a.l;
    return fast_two_sum(s.h, s.l);
}
SAGE_HD dd dd_mul_d(dd a, double b) {  // relative error <= 2^-104
    dd p = two_prod(a.h, b);
    p.l = crl_fma(a.l, b, p.l);
    return fast_two_sum(p.h, p.l);
}

SAGE_HD uint64_t crl_bits(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint64_t)__double_as_longlong(x);
#else
    uint64_t u;
    memcpy(&u, &x, 8);
    return u;
#endif
}
SAGE_HD double crl_from_bits(uint64_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __longlong_as_double((long long)u);
#else
    double x;
    memcpy(&x, &u, 8);
    return x;
#endif
}

// the accurate phase: ln x = e ln2 + lc + log1p(r) to a relative 2^-98
SAGE_HD double cr_log_accurate(const double e, const CrLogCell& t, const double r) {
    const double INV_D[10] = {SAGE_CRLOG_INV_D};    // (-1)^(n+1) / n, n = 9..18
    const dd INV_DD[7] = {SAGE_CRLOG_INV_DD};       // n = 2..8
    // tail: sum_{n=9..18} (-1)^(n+1) r^(n-9) / n in double (its weight is r^8 <= 2^-56 of the result)
    double tail = INV_D[9];
    for (int i = 8; i >= 0; i--) tail = crl_fma(tail, r, INV_D[i]);
    // Horner in double-double: acc = -1/8 + r tail; acc = 1/7 + r acc; ...; acc = -1/2 + r acc; then r + r^2 acc
    dd acc = dd_add_d(INV_DD[6], r * tail);
    for (int i = 5; i >= 0; i--) acc = dd_add(INV_DD[i], dd_mul_d(acc, r));
    acc = dd_mul_d(acc, r);          // r (-1/2 + ...)
    acc = dd_mul_d(acc, r);          // r^2 (-1/2 + ...)
    acc = dd_add_d(acc, r);          // log1p(r): |acc| <= 2^-15, r exact — (two_sum inside: no ordering assumption)
    // e ln2 as three exact / nearly exact pieces, largest first
    dd s = dd{e * SAGE_CRLOG_LN2_HI, 0.0};                 // exact (42-bit constant, |e| < 2^11)
    s = dd_add(s, dd{t.lh, t.ll});
    s = dd_add(s, two_prod(e, SAGE_CRLOG_LN2_MID));
    s = dd_add(s, acc);
    s = dd_add_d(s, e * SAGE_CRLOG_LN2_LO);
    return s.h + s.l;
}

// What both phases start from: x = 2^e m decomposed, the cell's constants, the exact r.
struct CrLogArg {
    double e, r;
    CrLogCell t;
    double special;  // the result when x is not a positive finite number (else 0)
    bool is_special;
};
SAGE_HD CrLogArg cr_log_reduce(double x) {
    CrLogArg a;
    a.is_special = false;
    a.special = 0.0;
    uint64_t u = crl_bits(x);
    int e_adj = 0;
    if (!(u - 0x0010000000000000ull < 0x7FE0000000000000ull)) {  // zero, subnormal, negative, inf, nan
        if (x != x || x == 0.0 || x < 0.0 || u == 0x7FF0000000000000ull) {
            a.is_special = true;
            a.special = x != x ? x : x == 0.0 ? -__builtin_huge_val() : x < 0.0 ? __builtin_nan("") : x;
            x = 1.0;  // (the arithmetic below then runs on a harmless argument)
            u = crl_bits(x);
        } else {
            x *= 0x1p54;  // subnormal
            u = crl_bits(x);
            e_adj = -54;
        }
    }
    const uint32_t k = (uint32_t)(u >> 45) & 127u;
    const int eb = (int)(u >> 52) - 1023 + e_adj + (k >= (uint32_t)SAGE_CRLOG_SPLIT ? 1 : 0);
    const double m = crl_from_bits((u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);
    a.t = CRLOG_TABLE[k];
    a.r = crl_fma(m, a.t.c, -1.0);  // exact
    a.e = (double)eb;
    return a;
}

// The fast phase: ln x as an unevaluated sum y.h + y.l with a relative error < 2^-81 (the bound used is 2^-79):
//   the leading sum e LN2_HI + lc.h + r - r^2/2 + r^3/3 - r^4/4 exactly as a chain of TwoSums (r^2, r^3, r^4 as exact products,
//   1/3 as a double-double), the series from r^5 on in double (its rounding error is 2^-52 r^5/5 <= 2^-82 of the result: the
//   result is at least 2^-7.01 unless it is log1p(r) itself, where the same ratio is 2^-52 r^4/5), truncated after r^13/13
//   (r^14/14 <= 2^-94).  Everything of the order 2^-53 of the result and below is summed in plain double: its rounding is 2^-106.
// `decided`: both ends of the error interval round to the same double — the correctly rounded result.  Undecided: ~2^-25 of
// the arguments.
SAGE_HD double cr_log_fast(const CrLogArg& a, bool& decided) {
    // (written to keep few values alive at a time — the callers are kernels at the edge of their register budget: the small terms
    // are folded into ONE running sum `low` as they appear; they are all below 2^-36 of the result, so the order of these
    // additions is immaterial at the 2^-89 level)
    const double e = a.e, r = a.r;
    const dd r2 = two_prod(r, r);                            // r^2 exactly
    double r3h = r2.h * r, r3l = crl_fma(r2.h, r, -r3h);     // r^3 = r2.h r + r2.l r
    r3l = crl_fma(r2.l, r, r3l);
    const double uh = r3h * 0x1.5555555555555p-2;            // r^3 / 3, 1/3 = 0x1.5555555555555p-2 + 0x1.5555555555555p-56
    double low = crl_fma(r3h, 0x1.5555555555555p-2, -uh);
    low = crl_fma(r3h, 0x1.5555555555555p-56, crl_fma(r3l, 0x1.5555555555555p-2, low));
    low = crl_fma(-0.5, r2.l, low);
    const double r4h = r2.h * r2.h;                          // r^4 = r2.h^2 + 2 r2.h r2.l (+ r2.l^2 <= 2^-106 r^4)
    low = crl_fma(-0.25, crl_fma(2.0 * r2.h, r2.l, crl_fma(r2.h, r2.h, -r4h)), low);
    double q = 0x1.3b13b13b13b14p-4;                         // 1/13
    q = crl_fma(q, r, -0x1.5555555555555p-4);                // -1/12
    q = crl_fma(q, r, 0x1.745d1745d1746p-4);                 // 1/11
    q = crl_fma(q, r, -0x1.999999999999ap-4);                // -1/10
    q = crl_fma(q, r, 0x1.c71c71c71c71cp-4);                 // 1/9
    q = crl_fma(q, r, -0.125);
    q = crl_fma(q, r, 0x1.2492492492492p-3);                 // 1/7
    q = crl_fma(q, r, -0x1.5555555555555p-3);                // -1/6
    q = crl_fma(q, r, 0.2);
    low = crl_fma(q, r4h * r, low);
    low += crl_fma(e, SAGE_CRLOG_LN2_MID, a.t.ll);
    dd s = two_sum(e * SAGE_CRLOG_LN2_HI, a.t.lh);           // (e * LN2_HI is exact)
    low += s.l;
    s = two_sum(s.h, r);
    low += s.l;
    s = two_sum(s.h, -0.5 * r2.h);
    low += s.l;
    s = two_sum(s.h, uh);
    low += s.l;
    s = two_sum(s.h, -0.25 * r4h);
    low += s.l;
    const dd y = fast_two_sum(s.h, low);
    const double err = __builtin_fabs(y.h) * 0x1p-79;
    const double lo = y.h + (y.l - err), hi = y.h + (y.l + err);
    decided = lo == hi;
    return lo;
}

// mode: 0 = two phases (production), 1 = accurate phase only, 2 = fast phase only (returns NaN where it cannot decide) — tests
template <int MODE = 0>
SAGE_HD double cr_log(double x) {
    const CrLogArg a = cr_log_reduce(x);
    if (a.is_special) return a.special;
    if (MODE == 1) return cr_log_accurate(a.e, a.t, a.r);
    bool decided;
    const double y = cr_log_fast(a, decided);
    if (decided) return y;
    if (MODE == 2) return __builtin_nan("");
    return cr_log_accurate(a.e, a.t, a.r);
}

}  // namespace sagecore
