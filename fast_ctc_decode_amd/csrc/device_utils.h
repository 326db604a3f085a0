// device_utils.h -- wave64 helpers for the gfx950 search kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fcd {

constexpr int kWave = 64;  // CDNA wavefront width; hard-coded on purpose

__device__ __forceinline__ uint64_t lanemask_lt() {
    return (1ull << (threadIdx.x & 63)) - 1ull;
}

__device__ __forceinline__ int popc64(uint64_t m) { return __builtin_popcountll(m); }

// wave-wide vote straight from an i1 (no 0/1 materialisation + compare as __ballot(int) does)
__device__ __forceinline__ uint64_t ballot(bool pred) { return __builtin_amdgcn_ballot_w64(pred); }

// Posterior element types (include/fcd.h: fcd_batch.dtype).  Basecaller networks emit half precision; the
// reference forces a host float32 copy (src/lib.rs:182,325).  Both 16-bit formats convert to binary32 EXACTLY, so
// a search on half-precision input is the reference's search on the upcast matrix, without a separate upcast pass.
enum { kF32 = 0, kF16 = 1, kBF16 = 2 };

#ifndef FCD_F16_TO_F32  // (tests/hipemu predefines a software conversion: its host compiler has no _Float16)
#define FCD_F16_TO_F32(h) ((float)__builtin_bit_cast(_Float16, (uint16_t)(h)))  // v_cvt_f32_f16: exact, subnormals included
#endif
__device__ __forceinline__ float f16_bits_to_f32(uint16_t h) { return FCD_F16_TO_F32(h); }
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

// the address of element `offset` of a posterior array (a read's first element), still typed as float *
__device__ __forceinline__ const float *post_at(const float *base, int64_t offset, int dtype) {
    return reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + offset * (dtype == kF32 ? 4 : 2));
}

// element `idx` of a posterior array of type `dtype` (wave-uniform: a scalar branch), as binary32
__device__ __forceinline__ float load_post(const float *base, int64_t idx, int dtype) {
    if (dtype == kF32) return base[idx];
    const uint16_t h = reinterpret_cast<const uint16_t *>(base)[idx];
    return dtype == kF16 ? f16_bits_to_f32(h) : bf16_bits_to_f32(h);
}

// Sort key for the prune step: descending probability, ties -> ascending node index
// (src/search.rs:245 stable sort by node followed by :262-269 sort by probability; see
// SURVEY.md 8a A4).  Larger key == earlier in the beam.  prob must not be NaN.
__device__ __forceinline__ uint64_t make_key(float prob, int node) {
    float p = prob + 0.0f;  // -0.0 -> +0.0 so that equal floats give equal keys
    uint32_t u = __float_as_uint(p);
    u ^= (u & 0x80000000u) ? 0xFFFFFFFFu : 0x80000000u;  // total order on non-NaN floats
    uint32_t lo = 0x7FFFFFFFu - (uint32_t)node;           // node >= -1; smaller node -> larger lo
    return ((uint64_t)u << 32) | lo;
}

// Loads that must observe this wave's own earlier global stores made from other lanes:
// served from L2 (sc1), bypassing the per-CU L1 (MI355X_MICROARCH.md, visibility table).
__device__ __forceinline__ int32_t load_i32_l2(const int32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Exact-rank building block: r_j += (k_j > key) for four comparands at once.  The compiler's own code for
// `rank += (k > key)` funnels every compare through VCC (v_cmp -> s_nop -> v_cndmask / v_addc), one long
// dependent chain; here the four compares land in four SGPR pairs and feed four independent accumulators,
// so consecutive instructions never wait on each other (each consumer sits three instructions behind its
// producer, which also covers the VALU-writes-SGPR -> VALU-reads-it wait states).
// (tests/hipemu predefines FCD_RANK4 in plain C++.)
#ifndef FCD_RANK4
#define FCD_RANK4(key, ka, kb, kc, kd, r0, r1, r2, r3)                                         \
    do {                                                                                       \
        uint64_t m0__, m1__, m2__, m3__;                                                       \
        asm("v_cmp_gt_u64_e64 %4, %9, %8\n\t"                                                  \
            "v_cmp_gt_u64_e64 %5, %10, %8\n\t"                                                 \
            "v_cmp_gt_u64_e64 %6, %11, %8\n\t"                                                 \
            "v_cmp_gt_u64_e64 %7, %12, %8\n\t"                                                 \
            "v_addc_co_u32_e64 %0, vcc, 0, %0, %4\n\t"                                         \
            "v_addc_co_u32_e64 %1, vcc, 0, %1, %5\n\t"                                         \
            "v_addc_co_u32_e64 %2, vcc, 0, %2, %6\n\t"                                         \
            "v_addc_co_u32_e64 %3, vcc, 0, %3, %7"                                              \
            : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "=&s"(m0__), "=&s"(m1__), "=&s"(m2__), "=&s"(m3__) \
            : "v"(key), "v"(ka), "v"(kb), "v"(kc), "v"(kd)                                     \
            : "vcc");                                                                          \
    } while (0)
#endif

// First block of a rank: the four accumulators START here (r_j = (k_j > key)), read off a shared zero register
// instead of being cleared one by one.
#ifndef FCD_RANK4_FIRST
#define FCD_RANK4_FIRST(key, ka, kb, kc, kd, r0, r1, r2, r3)                                   \
    do {                                                                                       \
        uint64_t m0__, m1__, m2__, m3__;                                                       \
        const int zero__ = 0;                                                                  \
        asm("v_cmp_gt_u64_e64 %4, %9, %8\n\t"                                                  \
            "v_cmp_gt_u64_e64 %5, %10, %8\n\t"                                                 \
            "v_cmp_gt_u64_e64 %6, %11, %8\n\t"                                                 \
            "v_cmp_gt_u64_e64 %7, %12, %8\n\t"                                                 \
            "v_addc_co_u32_e64 %0, vcc, 0, %13, %4\n\t"                                        \
            "v_addc_co_u32_e64 %1, vcc, 0, %13, %5\n\t"                                        \
            "v_addc_co_u32_e64 %2, vcc, 0, %13, %6\n\t"                                        \
            "v_addc_co_u32_e64 %3, vcc, 0, %13, %7"                                             \
            : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&s"(m0__), "=&s"(m1__), "=&s"(m2__), "=&s"(m3__) \
            : "v"(key), "v"(ka), "v"(kb), "v"(kc), "v"(kd), "v"(zero__)                        \
            : "vcc");                                                                          \
    } while (0)
#endif

// Makes a value opaque to the optimiser and pins it in vector registers (the duplex kernel's coefficient table).
// (tests/hipemu predefines FCD_OPAQUE_V as a no-op.)
#ifndef FCD_OPAQUE_V
#define FCD_OPAQUE_V(x) asm volatile("" : "+v"(x))
#endif

// Cycle stamp for the instrumented (PROF) kernel instantiations: reads the shader clock once every input
// the stamped block produced (`dep`) has arrived, and makes `dep` opaque so that nothing consuming it is
// scheduled above the stamp.  (tests/hipemu predefines FCD_STAMP as a no-op: there is no clock to read.)
#ifndef FCD_STAMP
#define FCD_STAMP(t64, dep) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t64), "+v"(dep) : : "memory")
#endif

}  // namespace fcd
