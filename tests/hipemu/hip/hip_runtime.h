// hip_runtime.h (tests/hipemu) -- TEST INFRASTRUCTURE ONLY.
//
// A lockstep wave64 emulator: the product's HIP kernels (fast_ctc_decode_amd/csrc/*.hip) are compiled
// UNCHANGED with g++ against this header instead of the ROCm one (tests/hipemu/build.py), so that the
// kernels' logic -- cross-lane traffic, ballots, DPP scans, LDS protocols, tree arena handling -- can be
// checked against the oracle on a machine without a GPU and debugged with ordinary host tools.
// Every work-item is a fibre; fibres of a wavefront meet at every cross-lane operation
// (ds_bpermute / ds_permute / ballot / readlane / DPP / shuffles / wave barrier), work-items of a
// block meet at __syncthreads().  Nothing here is part of, or reachable from, the product: the
// package loads libfcd_hip.so only, and only tests/ build or load the emulated library.
//
// What the emulation does NOT show: timing, register pressure, memory-model races that lockstep
// execution hides.  The -m gpu tests remain the parity gate.
#pragma once

#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FCD_OPAQUE_V(x) ((void)0)        // csrc/device_utils.h: register pinning of the duplex coefficient table
// csrc/device_utils.h: binary16 -> binary32 without _Float16 (g++ 11), exact incl. subnormals, infinities, NaNs
static inline float hipemu_f16_to_f32(unsigned short h) {
    const unsigned sign = (unsigned)(h & 0x8000u) << 16, ex = (h >> 10) & 0x1Fu, man = h & 0x3FFu;
    unsigned bits;
    if (ex == 0) {
        if (man == 0) {
            bits = sign;
        } else {  // subnormal: normalise
            int e = -1;
            unsigned m = man;
            do {
                ++e;
                m <<= 1;
            } while (!(m & 0x400u));
            bits = sign | (unsigned)(127 - 15 - e) << 23 | (m & 0x3FFu) << 13;
        }
    } else if (ex == 31) {
        bits = sign | 0x7F800000u | man << 13;
    } else {
        bits = sign | (ex + 127 - 15) << 23 | man << 13;
    }
    float f;
    __builtin_memcpy(&f, &bits, 4);
    return f;
}
#define FCD_F16_TO_F32(h) hipemu_f16_to_f32((unsigned short)(h))
#define FCD_STAMP(t64, dep) ((t64) = 0)  // csrc/device_utils.h: cycle stamps of the PROF instantiations
// csrc/device_utils.h: four compare-and-count steps (hand-scheduled VALU on the GPU)
#define FCD_RANK4(key, ka, kb, kc, kd, r0, r1, r2, r3) \
    do { (r0) += (ka) > (key); (r1) += (kb) > (key); (r2) += (kc) > (key); (r3) += (kd) > (key); } while (0)
#define FCD_RANK4_FIRST(key, ka, kb, kc, kd, r0, r1, r2, r3) \
    do { (r0) = (ka) > (key); (r1) = (kb) > (key); (r2) = (kc) > (key); (r3) = (kd) > (key); } while (0)

// ---- qualifiers -------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3

// ---- vector types -----------------------------------------------------------------------------
struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct __attribute__((aligned(8))) int2 { int x, y; };
struct __attribute__((aligned(16))) int4 { int x, y, z, w; };
struct __attribute__((aligned(8))) uint2 { unsigned x, y; };
struct __attribute__((aligned(16))) uint4 { unsigned x, y, z, w; };
struct __attribute__((aligned(8))) float2 { float x, y; };
struct __attribute__((aligned(16))) float4 { float x, y, z, w; };
struct __attribute__((aligned(16))) ulonglong2 { unsigned long long x, y; };
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

// ---- runtime API (host side of the C ABI) ---------------------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorUnknown = 999 };
typedef struct hipemu_stream *hipStream_t;
typedef struct hipemu_event *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1 };

hipError_t hipGetDeviceCount(int *n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int *d);
hipError_t hipMalloc(void **p, size_t n);
hipError_t hipFree(void *p);
constexpr unsigned hipHostMallocDefault = 0;
hipError_t hipHostMalloc(void **p, size_t n, unsigned flags);
hipError_t hipHostFree(void *p);
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b);
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind k, hipStream_t s);
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind k);
hipError_t hipMemset(void *dst, int v, size_t n);
// `__device__` variables are plain globals here (shared by every emulated device)
#define HIP_SYMBOL(x) (&(x))
inline hipError_t hipMemcpyToSymbol(void *symbol, const void *src, size_t n) {
    memcpy(symbol, src, n);
    return hipSuccess;
}
inline hipError_t hipMemcpyFromSymbol(void *dst, const void *symbol, size_t n) {
    memcpy(dst, symbol, n);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *dst, int v, size_t n, hipStream_t s);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipDeviceSynchronize();
hipError_t hipEventCreate(hipEvent_t *e);
#define hipEventDisableTiming 2u
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned flags, int priority);
hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
hipError_t hipGetLastError();
const char *hipGetErrorString(hipError_t e);

// ---- the scheduler ------------------------------------------------------------------------------
namespace hipemu {

struct Fiber;
extern Fiber *g_cur;  // the work-item that is running

struct Ids {
    dim3 thread, block, bdim, gdim;
};
const Ids &ids();
void *dyn_lds();  // dynamic LDS of the running block (16-byte aligned, zeroed per launch only)

// cross-lane primitives: `tag` names the operation so that a divergent wavefront (lanes meeting at
// different operations) is reported instead of silently mis-paired
enum Tag {
    T_BPERMUTE = 1, T_PERMUTE, T_BALLOT, T_READLANE, T_READFIRST, T_DPP, T_SHFL, T_SHFL_UP, T_SHFL_DOWN,
    T_SHFL_XOR, T_WAVE_BARRIER
};
struct Xchg {
    uint32_t val[64];
    uint32_t arg[64];
    uint64_t alive;  // lanes of the wavefront that have not returned from the kernel
};
// every live lane of the wavefront deposits (val, arg) and resumes once all have arrived; returns the
// snapshot of what they deposited
const Xchg &wave_exchange(int tag, uint32_t val, uint32_t arg);
void block_barrier();
int lane_id();

void launch_impl(dim3 grid, dim3 block, size_t lds_bytes, void (*tramp)(void *), void *closure);
void check_launch_stream(hipStream_t s);  // aborts when the stream belongs to another device than the current one

template <class F>
void launch(dim3 grid, dim3 block, size_t lds_bytes, F f) {
    launch_impl(grid, block, lds_bytes, [](void *c) { (*static_cast<F *>(c))(); }, &f);
}

}  // namespace hipemu

#define threadIdx (hipemu::ids().thread)
#define blockIdx (hipemu::ids().block)
#define blockDim (hipemu::ids().bdim)
#define gridDim (hipemu::ids().gdim)
#define warpSize 64

#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...) \
    (hipemu::check_launch_stream(stream), hipemu::launch((grid), (block), (size_t)(lds), [=]() { kern(__VA_ARGS__); }))

// ---- device intrinsics ----------------------------------------------------------------------------
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }

static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline long long min(long long a, long long b) { return a < b ? a : b; }
static inline long long max(long long a, long long b) { return a > b ? a : b; }
static inline long min(long a, long b) { return a < b ? a : b; }
static inline long max(long a, long b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }

static inline void __syncthreads() { hipemu::block_barrier(); }

static inline int __builtin_amdgcn_ds_bpermute(int addr, int v) {
    const hipemu::Xchg &x = hipemu::wave_exchange(hipemu::T_BPERMUTE, (uint32_t)v, (uint32_t)addr);
    const int src = (addr >> 2) & 63;
    return ((x.alive >> src) & 1) ? (int)x.val[src] : 0;
}
static inline int __builtin_amdgcn_ds_permute(int addr, int v) {
    const hipemu::Xchg &x = hipemu::wave_exchange(hipemu::T_PERMUTE, (uint32_t)v, (uint32_t)addr);
    const int me = hipemu::lane_id();
    int out = 0;  // a lane nobody writes to receives 0
    for (int l = 0; l < 64; ++l)
        if (((x.alive >> l) & 1) && (int)((x.arg[l] >> 2) & 63) == me) out = (int)x.val[l];
    return out;
}
static inline uint64_t __builtin_amdgcn_ballot_w64(bool p) {
    const hipemu::Xchg &x = hipemu::wave_exchange(hipemu::T_BALLOT, p ? 1u : 0u, 0);
    uint64_t m = 0;
    for (int l = 0; l < 64; ++l)
        if (((x.alive >> l) & 1) && x.val[l]) m |= 1ull << l;
    return m;
}
static inline uint64_t __ballot(int p) { return __builtin_amdgcn_ballot_w64(p != 0); }
static inline int __builtin_amdgcn_readlane(int v, int l) {
    const hipemu::Xchg &x = hipemu::wave_exchange(hipemu::T_READLANE, (uint32_t)v, (uint32_t)l);
    return (int)x.val[l & 63];
}
static inline int __builtin_amdgcn_readfirstlane(int v) {
    const hipemu::Xchg &x = hipemu::wave_exchange(hipemu::T_READFIRST, (uint32_t)v, 0);
    return (int)x.val[__builtin_ctzll(x.alive)];
}
static inline void __builtin_amdgcn_wave_barrier() { (void)hipemu::wave_exchange(hipemu::T_WAVE_BARRIER, 0, 0); }
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))

static inline int __shfl(int v, int src) {
    const hipemu::Xchg &x = hipemu::wave_exchange(hipemu::T_SHFL, (uint32_t)v, (uint32_t)src);
    return (int)x.val[src & 63];
}
static inline int __shfl_xor(int v, int m) {
    const hipemu::Xchg &x = hipemu::wave_exchange(hipemu::T_SHFL_XOR, (uint32_t)v, (uint32_t)m);
    return (int)x.val[(hipemu::lane_id() ^ m) & 63];
}
static inline int __shfl_up(int v, int d) {
    const hipemu::Xchg &x = hipemu::wave_exchange(hipemu::T_SHFL_UP, (uint32_t)v, (uint32_t)d);
    const int me = hipemu::lane_id();
    return me - d >= 0 ? (int)x.val[me - d] : v;
}
static inline int __shfl_down(int v, int d) {
    const hipemu::Xchg &x = hipemu::wave_exchange(hipemu::T_SHFL_DOWN, (uint32_t)v, (uint32_t)d);
    const int me = hipemu::lane_id();
    return me + d < 64 ? (int)x.val[me + d] : v;
}
static inline float __shfl(float v, int s) { return __int_as_float(__shfl(__float_as_int(v), s)); }
static inline float __shfl_xor(float v, int m) { return __int_as_float(__shfl_xor(__float_as_int(v), m)); }
static inline float __shfl_up(float v, int d) { return __int_as_float(__shfl_up(__float_as_int(v), d)); }
static inline float __shfl_down(float v, int d) { return __int_as_float(__shfl_down(__float_as_int(v), d)); }

// v_mov_b32 with a DPP control (ISA: DPP_CTRL values), row_mask / bank_mask gate the WRITE
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    const hipemu::Xchg &x = hipemu::wave_exchange(hipemu::T_DPP, (uint32_t)src, (uint32_t)ctrl);
    const int me = hipemu::lane_id();
    const int row = me >> 4, in_row = me & 15;
    if (!((row_mask >> row) & 1) || !((bank_mask >> (in_row >> 2)) & 1)) return old;
    int s = -1;  // source lane, -1 = out of range
    if (ctrl >= 0x000 && ctrl <= 0x0FF) s = (me & ~3) | ((ctrl >> (2 * (me & 3))) & 3);          // quad_perm
    else if (ctrl >= 0x101 && ctrl <= 0x10F) s = in_row + (ctrl & 15) <= 15 ? me + (ctrl & 15) : -1;  // row_shl
    else if (ctrl >= 0x111 && ctrl <= 0x11F) s = in_row - (ctrl & 15) >= 0 ? me - (ctrl & 15) : -1;   // row_shr
    else if (ctrl >= 0x121 && ctrl <= 0x12F) s = (me & ~15) | ((in_row - (ctrl & 15)) & 15);         // row_ror
    else if (ctrl == 0x130) s = me + 1 <= 63 ? me + 1 : -1;  // wave_shl:1
    else if (ctrl == 0x134) s = (me + 1) & 63;               // wave_rol:1
    else if (ctrl == 0x138) s = me - 1 >= 0 ? me - 1 : -1;   // wave_shr:1
    else if (ctrl == 0x13C) s = (me - 1) & 63;               // wave_ror:1
    else if (ctrl == 0x140) s = (me & ~15) | (15 - in_row);  // row_mirror
    else if (ctrl == 0x141) s = (me & ~7) | (7 - (me & 7));  // row_half_mirror
    else if (ctrl == 0x142) s = row >= 1 ? (row - 1) * 16 + 15 : -1;  // row_bcast:15
    else if (ctrl == 0x143) s = row >= 2 ? 31 : -1;                   // row_bcast:31
    else abort();
    if (s < 0 || !((x.alive >> s) & 1)) return bound_ctrl ? 0 : old;
    return (int)x.val[s];
}

// v_perm_b32: byte select from {src0 (bytes 7..4), src1 (bytes 3..0)}
static inline uint32_t __builtin_amdgcn_perm(uint32_t s0, uint32_t s1, uint32_t sel) {
    const uint64_t both = ((uint64_t)s0 << 32) | s1;
    uint32_t out = 0;
    for (int i = 0; i < 4; ++i) {
        const uint32_t c = (sel >> (8 * i)) & 0xFF;
        uint32_t b;
        if (c <= 7) b = (uint32_t)(both >> (8 * c)) & 0xFF;
        else if (c == 12) b = 0x00;
        else if (c >= 13) b = 0xFF;
        else b = ((both >> (16 * (c - 8) + 15)) & 1) ? 0xFF : 0x00;  // sign of a 16-bit half
        out |= b << (8 * i);
    }
    return out;
}

template <class T> static inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
template <class T> static inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <class T> static inline T atomicExch(T *p, T v) { T o = *p; *p = v; return o; }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
