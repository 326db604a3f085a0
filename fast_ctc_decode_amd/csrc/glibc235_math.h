// glibc235_math.h -- expf / logf / log1pf as glibc 2.35 computes them on x86-64, bit for bit.
//
// The reference's logsumexp arithmetic (`LogSpace`, /root/reference/src/duplex.rs:17,25,50, built without the default
// `fastexp` feature) is whatever `expf` / `logf` / `log1pf` the process links: Rust's f32::exp / ln / ln_1p call the C
// library.  The kernels' default definition is platform-independent (correctly rounded, logadd_fast.h); this header adds
// the one concrete libm the oracle can call in the build image -- glibc 2.35 (Ubuntu 22.04):
//   expf, logf   sysdeps/ieee754/flt-32/e_expf.c, e_logf.c (Szabolcs Nagy's table-driven routines: a 32-entry 2^(i/32)
//                table / a 16-entry {1/c, log c} table, polynomials evaluated in binary64, one rounding to binary32),
//                in the form the x86-64 multiarch build selects on a CPU with FMA (`__expf_fma`, `__logf_fma`: the same
//                source compiled -mfma, every a * b + c fused -- for expf that includes r = InvLn2N * x - kd);
//   log1pf       sysdeps/ieee754/flt-32/s_log1pf.c (the fdlibm routine in binary32 arithmetic, no multiarch variant).
// Restated from memory of those sources (neither glibc's source nor network access exists in the build image) and
// then checked against the image's own libm on EVERY binary32 argument: tools/verify/verify_glibc235.c -- expf, logf and
// log1pf each 4 294 967 296 arguments, 0 differences (profiles/r04_glibc235_verify.txt).  Without the fused r, expf
// differs on exactly two arguments (0x1.04845ep+5, -0x1.f8cbb2p+5): what a CPU without FMA would compute.
// Attribution: the algorithms and table constants are those of the GNU C Library 2.35 (LGPL-2.1-or-later).  expf and
// logf there are Arm's optimized routines by Szabolcs Nagy (Copyright (c) 2017-2018 Arm Ltd; MIT OR Apache-2.0 WITH
// LLVM-exception upstream, github.com/ARM-software/optimized-routines); log1pf descends from fdlibm's s_log1pf.c
// (Copyright (C) 1993 Sun Microsystems, Inc.: "Permission to use, copy, modify, and distribute this software is freely
// granted, provided that this notice is preserved").  No file was copied; what is restated is the arithmetic.
// Plain IEEE binary64 / binary32 arithmetic and fma(): the same code runs on the host (the verification, the threshold's
// logarithm in capi.hip) and on gfx950 (v_fma_f64, IEEE division).
#pragma once

#include <math.h>
#include <stdint.h>
#include <string.h>

#ifndef FCD_G235_FN
#if defined(__HIPCC__) || defined(FCD_HIPEMU)
#define FCD_G235_FN __host__ __device__ static inline
#else
#define FCD_G235_FN static inline
#endif
#endif

namespace fcd {
namespace g235 {

FCD_G235_FN uint32_t asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
FCD_G235_FN float asfloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
FCD_G235_FN uint64_t asuint64(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
FCD_G235_FN double asdouble(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }

FCD_G235_FN uint64_t exp2f_tab(int i) {
    const uint64_t T[32] = {
0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51,
0x3fef72b83c7d517b, 0x3fef54873168b9aa, 0x3fef387a6e756238, 0x3fef1e9df51fdee1,
0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d,
0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429, 0x3feea47eb03a5585,
0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74, 0x3feea11473eb0187, 0x3feea589994cce13,
0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d,
0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069,
0x3fef5818dcfba487, 0x3fef7c97337b9b5f, 0x3fefa4afa2a490da, 0x3fefd0765b6e4540,
    };
    return T[i];
}
FCD_G235_FN float expf235(float x) {
    const double InvLn2N = 0x1.71547652b82fep+0 * 32, SHIFT = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32, C2 = 0x1.62e42ff0c52d6p-1 / 32;
    double xd = (double)x;
    uint32_t abstop = (asuint(x) >> 20) & 0x7ff;
    if (abstop >= ((asuint(88.0f) >> 20) & 0x7ff)) {
        if (asuint(x) == asuint(-__builtin_huge_valf())) return 0.0f;
        if (abstop >= ((asuint(__builtin_huge_valf()) >> 20) & 0x7ff)) return x + x;
        if (x > 0x1.62e42ep6f) return __builtin_huge_valf();  /* overflow */
        if (x < -0x1.9fe368p6f) return 0.0f;     /* underflow (may_uflow) */
    }
    double z = InvLn2N * xd;
    double kd = z + SHIFT;
    uint64_t ki = asuint64(kd);
    kd -= SHIFT;
    const double r = fma(InvLn2N, xd, -kd);  // (fused, as in __expf_fma; see the header)
    uint64_t t = exp2f_tab((int)(ki % 32));
    t += ki << (52 - 5);
    double s = asdouble(t);
    z = fma(C0, r, C1);
    double r2 = r * r;
    double y = fma(C2, r, 1.0);
    y = fma(z, r2, y);
    y = y * s;
    return (float)y;
}
FCD_G235_FN void logf_tab(int i, double &invc, double &logc) {
    const double T[16][2] = {
  { 0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2 }, { 0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2 },
  { 0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2 }, { 0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3 },
  { 0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3 }, { 0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3 },
  { 0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4 }, { 0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4 },
  { 0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5 }, { 0x1p+0, 0x0p+0 },
  { 0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5 }, { 0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4 },
  { 0x1.b2036576afce6p-1, 0x1.526e57720db08p-3 }, { 0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3 },
  { 0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2 }, { 0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2 },
    };
    invc = T[i][0];
    logc = T[i][1];
}
FCD_G235_FN float logf235(float x) {
    const double Ln2 = 0x1.62e42fefa39efp-1, A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
    uint32_t ix = asuint(x);
    if (ix == 0x3f800000) return 0.0f;
    if (ix - 0x00800000 >= 0x7f800000 - 0x00800000) {
        if (ix * 2 == 0) return -__builtin_huge_valf();
        if (ix == 0x7f800000) return x;
        if ((ix & 0x80000000) || ix * 2 >= 0xff000000) return (x - x) / (x - x);
        ix = asuint(x * 0x1p23f);
        ix -= 23u << 23;
    }
    uint32_t tmp = ix - 0x3f330000;
    int i = (tmp >> (23 - 4)) % 16;
    int k = (int32_t)tmp >> 23;
    uint32_t iz = ix - (tmp & 0xff800000);  /* 0x1ff << 23 */
    double invc, logc;
    logf_tab(i, invc, logc);
    double z = (double)asfloat(iz);
    double r = fma(z, invc, -1.0);
    double y0 = fma((double)k, Ln2, logc);
    double r2 = r * r;
    double y = fma(A1, r, A2);
    y = fma(A0, r2, y);
    y = fma(y, r2, (y0 + r));
    return (float)y;
}
FCD_G235_FN float log1pf235(float x) {
    const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f, two25 = 3.355443200e+07f,
      Lp1 = 6.6666668653e-01f, Lp2 = 4.0000000596e-01f, Lp3 = 2.8571429849e-01f, Lp4 = 2.2222198546e-01f,
      Lp5 = 1.8183572590e-01f, Lp6 = 1.5313838422e-01f, Lp7 = 1.4798198640e-01f, zero = 0.0f;
    float hfsq, f = 0, c = 0, s, z, R, u;
    int32_t k, hx, hu = 0, ax;
    hx = (int32_t)asuint(x);
    ax = hx & 0x7fffffff;
    k = 1;
    if (hx < 0x3ed413d7) {
        if (ax >= 0x3f800000) {
            if (x == -1.0f) return -two25 / zero;
            return (x - x) / (x - x);
        }
        if (ax < 0x31000000) {
            if (ax < 0x24800000) return x;
            return x - x * x * 0.5f;
        }
        if (hx > 0 || hx <= ((int32_t)0xbe95f61f)) { k = 0; f = x; hu = 1; }
    }
    if (hx >= 0x7f800000) return x + x;
    if (k != 0) {
        if (hx < 0x5a000000) {
            u = 1.0f + x;
            hu = (int32_t)asuint(u);
            k = (hu >> 23) - 127;
            c = (k > 0) ? 1.0f - (u - x) : x - (u - 1.0f);
            c /= u;
        } else {
            u = x;
            hu = (int32_t)asuint(u);
            k = (hu >> 23) - 127;
            c = 0;
        }
        hu &= 0x007fffff;
        if (hu < 0x3504f7) {
            u = asfloat((uint32_t)hu | 0x3f800000);
        } else {
            k += 1;
            u = asfloat((uint32_t)hu | 0x3f000000);
            hu = (0x00800000 - hu) >> 2;
        }
        f = u - 1.0f;
    }
    hfsq = 0.5f * f * f;
    if (hu == 0) {
        if (f == zero) {
            if (k == 0) return zero;
            c += k * ln2_lo;
            return k * ln2_hi + c;
        }
        R = hfsq * (1.0f - 0.66666666666666666f * f);
        if (k == 0) return f - R;
        return k * ln2_hi - ((R - (k * ln2_lo + c)) - f);
    }
    s = f / (2.0f + f);
    z = s * s;
    R = z * (Lp1 + z * (Lp2 + z * (Lp3 + z * (Lp4 + z * (Lp5 + z * (Lp6 + z * Lp7))))));
    if (k == 0) return f - (hfsq - s * (hfsq + R));
    return k * ln2_hi - ((hfsq - (s * (hfsq + R) + (k * ln2_lo + c))) - f);
}

}  // namespace g235
}  // namespace fcd
