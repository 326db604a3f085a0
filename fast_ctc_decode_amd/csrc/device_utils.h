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

// Cycle stamp for the instrumented (PROF) kernel instantiations: reads the shader clock once every input
// the stamped block produced (`dep`) has arrived, and makes `dep` opaque so that nothing consuming it is
// scheduled above the stamp.  (tests/hipemu predefines FCD_STAMP as a no-op: there is no clock to read.)
#ifndef FCD_STAMP
#define FCD_STAMP(t64, dep) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t64), "+v"(dep) : : "memory")
#endif

}  // namespace fcd
