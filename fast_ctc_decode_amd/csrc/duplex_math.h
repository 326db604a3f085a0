// duplex_math.h -- LogSpace arithmetic (/root/reference/src/duplex.rs:7-80) and small wave64 helpers shared by the two
// duplex kernels (duplex.hip: any shape, beam in LDS; duplex_slots.hip: the slot-resident kernel for nslots <= 64).
// Everything here is per translation unit (anonymous namespace), exactly as it was inside duplex.hip.
#pragma once

#include "device_utils.h"
#include "fcd_internal.h"
#include "glibc235_math.h"
#include "logadd_fast.h"

namespace fcd {

namespace {

constexpr float kNegInf = -__builtin_huge_valf();

__device__ __forceinline__ float ln_cr(float x) { return (float)log((double)x); }

// The slow paths of LogSpace::add -- the library routines, taken for ~1e-6 of the arguments -- stay out of
// line: inlined into the window-building loop they cost it ~160 scalar-register spills and a third of its
// instructions, every iteration, for code that almost never runs.
__device__ __attribute__((noinline)) float exp_slow_f32(float x) { return (float)exp((double)x); }
__device__ __attribute__((noinline)) float log1p_slow_f32(float e) { return (float)log1p((double)e); }

template <int MODE>
__device__ __forceinline__ float ladd(float a, float b, const LogAddCoef &K) {
    // duplex.rs:42-63: operands ordered so that a NaN ends up in `big`
    float big, small;
    if (a <= b) {
        big = b;
        small = a;
    } else {
        big = a;
        small = b;
    }
    if (small == kNegInf) return big;
    if (MODE == FCD_LOGADD_MAX) return big + 0.0f;
    // FCD_LOGADD_LOGSUMEXP_GLIBC235: the same expression on glibc 2.35's expf / log1pf, bit for bit (glibc235_math.h)
    if (MODE == FCD_LOGADD_LOGSUMEXP_GLIBC235) return big + g235::log1pf235(g235::expf235(small - big));
    // big + ln_1p(exp(small - big)), exp and ln_1p each correctly rounded to f32.  Fast binary64
    // evaluations (logadd_fast.h, verified exhaustively on the host) with Ziv's rounding test; the
    // general-purpose library routines only run for the ~1e-6 of arguments that test rejects and
    // for exponentials with subnormal results.
    const float x = small - big;                      // <= 0, or NaN
    if (x < kExpZeroBelow) return big + 0.0f;         // exp -> +0, ln_1p(+0) = +0
    // x < -86: e = exp(x) <= 4.5e-38 and ln_1p(e) = e; adding it to a `big` of magnitude >= 2^-90
    // (half a unit in the last place >= 2^-115) cannot change `big`
    if (x < kExpFastMin && __builtin_fabsf(big) >= 8.0779356694631609e-28f) return big;
    const double ye = exp_fast((double)x, K);
    float e = (float)ye;
    if (!(x >= kExpFastMin) || round_to_f32_unsafe(ye)) e = exp_slow_f32(x);
    if (e < kLog1pIdentityBelow) return big + e;      // ln_1p(e) rounds to e below 2^-24
    const double yl = log1p_fast((double)e, K);
    float l = (float)yl;
    if (round_to_f32_unsafe(yl)) l = log1p_slow_f32(e);
    return big + l;
}

template <int MODE>
__device__ __forceinline__ float ladd(float a, float b) { return ladd<MODE>(a, b, logadd_coef()); }

// The same function for the window-building loop, where every lane of the wavefront calls it in lockstep: the
// shortcuts become selects, and ONE wave-wide test skips the transcendental part when no lane needs it (rows
// far from the alignment, where exp(small - big) is 0 for every new node) -- instead of a nest of
// exec-mask branches per row.  Operation for operation the values are those of ladd().
template <int MODE>
__device__ __forceinline__ float ladd_lockstep(float a, float b, const LogAddCoef &K) {
    const bool ab = a <= b;
    const float big = ab ? b : a, small = ab ? a : b;  // a NaN ends up in `big` or makes x NaN
    if (MODE == FCD_LOGADD_MAX) return small == kNegInf ? big : big + 0.0f;
    if (MODE == FCD_LOGADD_LOGSUMEXP_GLIBC235)  // (a parity mode: no fast paths, no votes)
        return small == kNegInf ? big : big + g235::log1pf235(g235::expf235(small - big));
    const float x = small - big;  // <= 0, or NaN
    // ladd()'s shortcuts, folded: the transcendental part is needed unless x < -86 -- and then it is still needed when
    // `big` is so small (below 2^-90) that exp(x) could show in the sum, which the slow exponential handles down to
    // x < -104 where it returns +0 and the sum is big + 0.  small = -inf returns `big` whatever x is (NaN for
    // -inf - -inf).
    const bool sc_inf = small == kNegInf;
    const bool full = (!(x < kExpFastMin) | (__builtin_fabsf(big) < 8.0779356694631609e-28f)) & !sc_inf;
    float res = big;
    if (ballot(full) != 0ull) {
        const float xs = full ? x : -1.0f;  // lanes that do not need it still run the arithmetic, on a tame argument
        // e = exp(xs) rounded to f32, carried as a binary64 value (it is ln_1p's argument); ln_1p's fast path covers
        // every normal e, so "ln_1p(e) = e below 2^-24" needs no select here.  (Bitwise |: nothing to skip.)
        const double ye = exp_fast((double)xs, K);
        double ed = round_to_f32_as_f64(ye);
        if ((int)!(xs >= kExpFastMin) | (int)round_to_f32_unsafe(ye)) ed = (double)exp_slow_f32(xs);
        const double yl = log1p_fast(ed, K);
        float l = (float)yl;
        // (e below 2^-126 -- only after exp's slow path -- by its exponent field: one 32-bit compare)
        if ((int)((uint32_t)(bits_of(ed) >> 32) < 0x38100000u) | (int)round_to_f32_unsafe(yl)) {
            const float e = (float)ed;  // exact
            l = e < kLog1pIdentityBelow ? e : log1p_slow_f32(e);
        }
        res = full ? big + l : res;
    }
    return res;
}

__device__ __forceinline__ float lmax(float self, float other) { return self < other ? other : self; }

// LogSpace-style maximum over the 64 lanes, delivered to every lane: the classic GCN DPP reduction (row shifts, then
// the two row broadcasts) ends in lane 63, which a readlane hands out -- no LDS round trips (six ds_bpermute steps cost
// ~800 cycles per call with one wavefront on the SIMD).  Operands are never NaN here (callers fold with lmax from -inf).
__device__ __forceinline__ float wave_lmax(float x) {
    float t = x;
#define FCD_DPP_LMAX(CTRL, RM)                                                                                          \
    t = lmax(t, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(t), __float_as_int(t), CTRL, RM, 0xf, false)));
    FCD_DPP_LMAX(0x111, 0xf)  // row_shr:1
    FCD_DPP_LMAX(0x112, 0xf)  // row_shr:2
    FCD_DPP_LMAX(0x114, 0xf)  // row_shr:4
    FCD_DPP_LMAX(0x118, 0xf)  // row_shr:8   -> lane 15 of every row holds the row's maximum
    FCD_DPP_LMAX(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
    FCD_DPP_LMAX(0x143, 0xc)  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the maximum
#undef FCD_DPP_LMAX
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 63));
}

// v_max_f32 as it is: through __builtin_fmaxf the compiler first "canonicalises" every operand it cannot prove free of
// signalling NaNs (v_max_f32 x, x, x) -- on the dependent chain.  Callers guarantee ordinary operands.
__device__ __forceinline__ float vmax_raw(float a, float b) {
#ifdef FCD_HIPEMU
    return a != a ? b : (b != b ? a : (a < b ? b : a));  // (the instruction's IEEE maxNum: a NaN operand is dropped)
#else
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#endif
}

// one window row {label, gap, sum} in one 12-byte store (rows are 12 bytes apart: 4-byte alignment is all there is)
struct __attribute__((packed, aligned(4))) Row3 {
    float lb, g, sm;
};
__device__ __forceinline__ void store_row(float *p, float lb, float g, float sm) {
    *reinterpret_cast<Row3 *>(p) = Row3{lb, g, sm};
}

__device__ __forceinline__ float load_f32_l2(const float *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int4 load_meta_l2(const int4 *p) {
    const int32_t *q = reinterpret_cast<const int32_t *>(p);
    return make_int4(load_i32_l2(q), load_i32_l2(q + 1), load_i32_l2(q + 2), load_i32_l2(q + 3));
}

}  // namespace

}  // namespace fcd
