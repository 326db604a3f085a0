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
// < 2^-47 relative), scaled by 2^k.
FCD_HD double exp_fast(double x) {
    const double k = rint(x * 1.4426950408889634074);
    double r = fma(k, -6.93147180369123816490e-01, x);  // ln2 split as in fdlibm: hi has 21 trailing zeros
    r = fma(k, -1.90821492927058770002e-10, r);
    double p = 1.0 / 39916800.0;
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)k);
}

// ln_1p(e) = 2 atanh(s), s = e / (2 + e) in (0, 1/3]: odd series through s^29 (truncation < 2^-52 relative).
FCD_HD double log1p_fast(double e) {
    const double s = e / (2.0 + e);
    const double z = s * s;
    double p = 1.0 / 29.0;
    p = fma(p, z, 1.0 / 27.0);
    p = fma(p, z, 1.0 / 25.0);
    p = fma(p, z, 1.0 / 23.0);
    p = fma(p, z, 1.0 / 21.0);
    p = fma(p, z, 1.0 / 19.0);
    p = fma(p, z, 1.0 / 17.0);
    p = fma(p, z, 1.0 / 15.0);
    p = fma(p, z, 1.0 / 13.0);
    p = fma(p, z, 1.0 / 11.0);
    p = fma(p, z, 1.0 / 9.0);
    p = fma(p, z, 1.0 / 7.0);
    p = fma(p, z, 1.0 / 5.0);
    p = fma(p, z, 1.0 / 3.0);
    p = fma(p, z, 1.0);
    return (2.0 * s) * p;
}

constexpr float kExpFastMin = -86.0f;           // below: exp's f32 result may be subnormal
constexpr float kExpZeroBelow = -104.0f;        // below: exp rounds to +0 (2^-150 = e^-103.97)
constexpr float kLog1pIdentityBelow = 5.9604644775390625e-08f;  // 2^-24: below, ln_1p(e) rounds to e

}  // namespace fcd
