// logadd_fast.h -- fast paths of the correctly rounded f32 exp / ln_1p behind LogSpace::add
// (/root/reference/src/duplex.rs:42-63), shared by the duplex kernels (device) and by the exhaustive
// host-side verifier tools/verify/verify_logadd.cpp (every operation below is an IEEE-754 binary64
// operation with one rounding -- fma, rint, ldexp, division -- so host and device compute the same bits).
//
// Definition being implemented: exp and ln_1p return the f32 nearest to the exact
// value.  The fast paths evaluate in binary64 with a relative error below 2^-46 and round once; when the
// binary64 value lies within 512 ulps (2^-43 relative) of an f32 rounding boundary the caller falls back
// to the slow path (Ziv's test).  Domains: exp x in [-86, 0] (normal f32 results), ln_1p e in [2^-24, 1].
#pragma once

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define FCD_HD __host__ __device__ __forceinline__
#else
#define FCD_HD static inline
#endif

namespace fcd {

FCD_HD uint64_t bits_of(double x) {
    uint64_t u;
    memcpy(&u, &x, sizeof u);
    return u;
}

// y: positive normal binary64 whose f32 rounding is a normal number.  True when y is so close to the
// midpoint of two adjacent f32 values that an error of 2^-43 (relative) could change the rounding.
FCD_HD bool round_to_f32_unsafe(double y) {
    const uint32_t dropped = (uint32_t)bits_of(y) & 0x1FFFFFFFu;  // the 29 mantissa bits f32 drops
    return (uint32_t)(dropped - (0x10000000u - 512u)) < 1024u;
}

// exp(x), x in [-86, 0]: x = k ln2 + r, |r| <= 0.3466, Taylor polynomial of degree 11 (truncation
// < 2^-47 relative), scaled by 2^k.  The polynomial is evaluated by Estrin's scheme: the searches run one
// wavefront per SIMD, where the DEPENDENT chain of a log-add is what a window row costs -- five levels of
// independent fused multiply-adds instead of eleven in a row (same operation count).
FCD_HD double exp_fast(double x) {
    const double k = rint(x * 1.4426950408889634074);
    double r = fma(k, -6.93147180369123816490e-01, x);  // ln2 split as in fdlibm: hi has 21 trailing zeros
    r = fma(k, -1.90821492927058770002e-10, r);
    const double r2 = r * r, r4 = r2 * r2, r8 = r4 * r4;
    const double p01 = 1.0 + r;                                   // c0 + c1 r
    const double p23 = fma(1.0 / 6.0, r, 0.5);                    // c2 + c3 r
    const double p45 = fma(1.0 / 120.0, r, 1.0 / 24.0);
    const double p67 = fma(1.0 / 5040.0, r, 1.0 / 720.0);
    const double p89 = fma(1.0 / 362880.0, r, 1.0 / 40320.0);
    const double pab = fma(1.0 / 39916800.0, r, 1.0 / 3628800.0);
    const double q03 = fma(p23, r2, p01);
    const double q47 = fma(p67, r2, p45);
    const double q8b = fma(pab, r2, p89);
    const double h07 = fma(q47, r4, q03);
    const double p = fma(q8b, r8, h07);
    return ldexp(p, (int)k);
}

// ln_1p(e) = 2 atanh(s), s = e / (2 + e) in (0, 1/3]: odd series through s^29 (truncation < 2^-52 relative),
// Estrin's scheme in z = s^2 (fifteen coefficients: four levels).
FCD_HD double log1p_fast(double e) {
    const double s = e / (2.0 + e);
    const double z = s * s, z2 = z * z, z4 = z2 * z2, z8 = z4 * z4;
    const double a0 = fma(1.0 / 3.0, z, 1.0);
    const double a1 = fma(1.0 / 7.0, z, 1.0 / 5.0);
    const double a2 = fma(1.0 / 11.0, z, 1.0 / 9.0);
    const double a3 = fma(1.0 / 15.0, z, 1.0 / 13.0);
    const double a4 = fma(1.0 / 19.0, z, 1.0 / 17.0);
    const double a5 = fma(1.0 / 23.0, z, 1.0 / 21.0);
    const double a6 = fma(1.0 / 27.0, z, 1.0 / 25.0);
    const double a7 = 1.0 / 29.0;
    const double b0 = fma(a1, z2, a0);
    const double b1 = fma(a3, z2, a2);
    const double b2 = fma(a5, z2, a4);
    const double b3 = fma(a7, z2, a6);
    const double c0 = fma(b1, z4, b0);
    const double c1 = fma(b3, z4, b2);
    const double p = fma(c1, z8, c0);
    return (2.0 * s) * p;
}

constexpr float kExpFastMin = -86.0f;           // below: exp's f32 result may be subnormal
constexpr float kExpZeroBelow = -104.0f;        // below: exp rounds to +0 (2^-150 = e^-103.97)
constexpr float kLog1pIdentityBelow = 5.9604644775390625e-08f;  // 2^-24: below, ln_1p(e) rounds to e

}  // namespace fcd
