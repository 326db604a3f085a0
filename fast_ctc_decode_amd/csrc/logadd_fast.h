// logadd_fast.h -- fast paths of the correctly rounded f32 exp / ln_1p behind LogSpace::add
// (/root/reference/src/duplex.rs:42-63), shared by the duplex kernels (device) and by the exhaustive
// host-side verifier tools/verify/verify_logadd.cpp (every operation below is an IEEE-754 binary64
// operation with one rounding -- fma, rint, ldexp, division -- so host and device compute the same bits).
//
// Definition being implemented: exp and ln_1p return the f32 nearest to the exact
// value.  The fast paths evaluate in binary64 with a relative error below 2^-46 and round once; when the
// binary64 value lies within 512 ulps (2^-43 relative) of an f32 rounding boundary the caller falls back
// to the slow path (Ziv's test).  Domains: exp x in [-86, 0] (normal f32 results), ln_1p e in [2^-126, 1].
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
FCD_HD double double_of(uint64_t u) {
    double x;
    memcpy(&x, &u, sizeof x);
    return x;
}

// y: positive normal binary64 whose f32 rounding is a normal number.  True when y is so close to the
// midpoint of two adjacent f32 values that an error of 2^-43 (relative) could change the rounding.
FCD_HD bool round_to_f32_unsafe(double y) {
    const uint32_t dropped = (uint32_t)bits_of(y) & 0x1FFFFFFFu;  // the 29 mantissa bits f32 drops
    return (uint32_t)(dropped - (0x10000000u - 512u)) < 1024u;
}

// y as above and NOT round_to_f32_unsafe: the f32 nearest to y, as a binary64 value, by integer arithmetic (add
// half a unit of the 29 dropped bits, clear them; a carry out of the mantissa lands in the exponent, as it should;
// an exact tie cannot occur, the test rejects everything near one).  Two full-rate integer operations where
// (double)(float)y is two quarter-rate conversions on the GPU.
FCD_HD double round_to_f32_as_f64(double y) { return double_of((bits_of(y) + 0x10000000ull) & ~0x1FFFFFFFull); }

// The coefficients as a value: the window-building loop of the duplex kernel keeps them in vector registers
// for its whole run (made opaque there, so the compiler neither re-materialises 64-bit literals through scalar
// registers on every row nor spills scalars to make room for them).
struct LogAddCoef {
    double log2e, ln2hi, ln2lo, magic;
    double e[12];   // 1/k!,  k = 0..11
    double a[15];   // 1/(2k+1), k = 0..14
    double two;
};

FCD_HD LogAddCoef logadd_coef() {
    LogAddCoef c;
    c.log2e = 1.4426950408889634074;
    c.ln2hi = -6.93147180369123816490e-01;  // ln2 split as in fdlibm: hi has 21 trailing zeros
    c.ln2lo = -1.90821492927058770002e-10;
    c.e[0] = 1.0; c.e[1] = 1.0; c.e[2] = 0.5; c.e[3] = 1.0 / 6.0; c.e[4] = 1.0 / 24.0; c.e[5] = 1.0 / 120.0;
    c.e[6] = 1.0 / 720.0; c.e[7] = 1.0 / 5040.0; c.e[8] = 1.0 / 40320.0; c.e[9] = 1.0 / 362880.0;
    c.e[10] = 1.0 / 3628800.0; c.e[11] = 1.0 / 39916800.0;
    for (int k = 0; k < 15; ++k) c.a[k] = 1.0 / (double)(2 * k + 1);
    c.two = 2.0;
    c.magic = 6755399441055744.0;  // 1.5 * 2^52: adding it leaves the nearest integer in the low mantissa bits
    return c;
}

// exp(x), x in [-86, 0]: x = k ln2 + r, |r| <= 0.3466, Taylor polynomial of degree 11 (truncation
// < 2^-47 relative), scaled by 2^k.  The polynomial is evaluated by Estrin's scheme: five levels of
// independent fused multiply-adds instead of eleven in a row (same operation count).  k -- the integer nearest
// to x log2(e) -- comes out of one fused multiply-add against 1.5 * 2^52 (value: t - magic; as an integer: the low
// word of t), and since k is in [-125, 0] and the polynomial in [0.70, 1.42] the result is a normal number:
// scaling by 2^k is an integer addition to the exponent field.  Five full-rate operations where rint, the
// double -> int conversion and ldexp are three quarter-rate ones on the GPU.  (Outside the domain the value is
// meaningless, not harmful: callers send x < -86 and NaN to the slow path.)
FCD_HD double exp_fast(double x, const LogAddCoef &c) {
    const double t = fma(x, c.log2e, c.magic);
    const double k = t - c.magic;
    double r = fma(k, c.ln2hi, x);
    r = fma(k, c.ln2lo, r);
    const double r2 = r * r, r4 = r2 * r2, r8 = r4 * r4;
    const double p01 = c.e[0] + r;                 // c0 + c1 r
    const double p23 = fma(c.e[3], r, c.e[2]);     // c2 + c3 r
    const double p45 = fma(c.e[5], r, c.e[4]);
    const double p67 = fma(c.e[7], r, c.e[6]);
    const double p89 = fma(c.e[9], r, c.e[8]);
    const double pab = fma(c.e[11], r, c.e[10]);
    const double q03 = fma(p23, r2, p01);
    const double q47 = fma(p67, r2, p45);
    const double q8b = fma(pab, r2, p89);
    const double h07 = fma(q47, r4, q03);
    const double p = fma(q8b, r8, h07);
    return double_of(bits_of(p) + ((uint64_t)((uint32_t)bits_of(t) << 20) << 32));
}
FCD_HD double exp_fast(double x) { return exp_fast(x, logadd_coef()); }

// ln_1p(e) = 2 atanh(s), s = e / (2 + e) in (0, 1/3]: odd series through s^29 (truncation < 2^-52 relative),
// Estrin's scheme in z = s^2 (fifteen coefficients: four levels).  Verified for every f32 e in [2^-126, 1] (below
// 2^-24 the correctly rounded result is e itself, and this evaluation delivers it: no special case needed).
FCD_HD double log1p_fast(double e, const LogAddCoef &c) {
#if defined(__HIP_DEVICE_COMPILE__)
    // On the GPU the quotient goes through the hardware reciprocal and two Newton steps: five dependent operations
    // where the IEEE division is eleven, on a chain whose LATENCY is what the duplex search waits for (DESIGN.md
    // 4.4).  The result may differ from the division's in the last bits -- far inside the 2^-43 the rounding test
    // below it tolerates -- so both land on the same f32 whenever the test trusts them; the host verifier keeps the
    // division, and the device variant is swept exhaustively on the device itself (fcd_logadd_sweep_dev,
    // tests/test_gpu_duplex.py::test_logadd_fast_paths_exhaustive_on_device).
    const double d = c.two + e;
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    const double s = e * r;
#else
    const double s = e / (c.two + e);
#endif
    const double z = s * s, z2 = z * z, z4 = z2 * z2, z8 = z4 * z4;
    const double a0 = fma(c.a[1], z, c.a[0]);
    const double a1 = fma(c.a[3], z, c.a[2]);
    const double a2 = fma(c.a[5], z, c.a[4]);
    const double a3 = fma(c.a[7], z, c.a[6]);
    const double a4 = fma(c.a[9], z, c.a[8]);
    const double a5 = fma(c.a[11], z, c.a[10]);
    const double a6 = fma(c.a[13], z, c.a[12]);
    const double a7 = c.a[14];
    const double b0 = fma(a1, z2, a0);
    const double b1 = fma(a3, z2, a2);
    const double b2 = fma(a5, z2, a4);
    const double b3 = fma(a7, z2, a6);
    const double c0 = fma(b1, z4, b0);
    const double c1 = fma(b3, z4, b2);
    const double p = fma(c1, z8, c0);
    return (c.two * s) * p;
}
FCD_HD double log1p_fast(double e) { return log1p_fast(e, logadd_coef()); }

constexpr float kExpFastMin = -86.0f;           // below: exp's f32 result may be subnormal
constexpr float kExpZeroBelow = -104.0f;        // below: exp rounds to +0 (2^-150 = e^-103.97)
constexpr float kLog1pIdentityBelow = 5.9604644775390625e-08f;  // 2^-24: below, ln_1p(e) rounds to e
constexpr float kLog1pFastMin = 1.17549435082228750797e-38f;     // 2^-126: smallest argument of the fast ln_1p

}  // namespace fcd
