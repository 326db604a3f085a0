// pack.hip -- compact wire format of a shard's decoded results for the ONE gather of the multi-GPU path
// (fast_ctc_decode_amd/dist.py).  The searches write fixed-stride rows (out_stride >= T
// entries per read) of which ~48 % are used at BASELINE config 2; shipping only the used prefix of every
// row halves the bytes on the xGMI link:
//
//   [ 0, 8)              u64  total = sum(out_len)
//   [ 8,16)              u32  n_reads, u32 path_bytes (2 or 4)
//   [16, 16+4n)          u32  out_len[n]
//   [16+4n, 16+8n)       i32  status[n]
//   [16+8n, +align4(total))   u8   labels of read 0, read 1, ... back to back
//   [.., + total*path_bytes)  u16/u32 path entries, same order
//   host result chunks only (hostjob.hip; never on the wire): the path region is absent when no path was asked for
//   (header word 3 = 0), and with header word 3 bit 8 set an f32 region of `total` quality values follows,
//   4-byte aligned
//
// fcd_result_offsets_dev: exclusive prefix sum of out_len (one workgroup, DPP-free shuffle scan).
// fcd_pack_results_dev / fcd_unpack_results_dev: one wavefront per read, coalesced 4-byte source loads.
#include "device_utils.h"
#include "fcd_internal.h"

namespace fcd {

namespace {

constexpr int kScanThreads = 1024;

// offsets[i] = sum of len[0..i), offsets[n] = total; len[i] is clamped to `stride`
__global__ __launch_bounds__(kScanThreads) void result_offsets_kernel(const uint32_t *len, int64_t n, int64_t stride,
                                                                      uint64_t *offsets) {
    __shared__ uint64_t s_part[kScanThreads / 64];
    __shared__ uint64_t s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += kScanThreads) {
        const int64_t i = base + tid;
        uint64_t v = 0;
        if (i < n) {
            const uint64_t l = len[i];
            v = l < (uint64_t)stride ? l : (uint64_t)stride;
        }
        // inclusive scan inside the wavefront (two 32-bit halves per shuffle)
        uint64_t incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)incl, o);
            const uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(incl >> 32), o);
            if (lane >= o) incl += ((uint64_t)hi << 32) | lo;
        }
        if (lane == 63) s_part[wave] = incl;
        __syncthreads();
        uint64_t before = s_carry;
        for (int w = 0; w < wave; ++w) before += s_part[w];
        if (i < n) offsets[i] = before + incl - v;
        __syncthreads();
        if (tid == kScanThreads - 1) s_carry = before + incl;
        __syncthreads();
    }
    if (tid == 0) offsets[n] = s_carry;
}

struct PackParams {
    const uint8_t *labels;
    const uint32_t *path;   // nullable
    const float *qual;      // nullable
    const uint32_t *out_len;
    const int32_t *status;
    int64_t stride;
    int64_t n;
    int path_bytes;
    const uint64_t *offsets;
    uint8_t *buf;
};

__device__ __forceinline__ size_t header_bytes(int64_t n) { return 16 + 8 * (size_t)n; }

__global__ __launch_bounds__(256) void pack_kernel(PackParams p) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= p.n) return;
    const uint64_t total = p.offsets[p.n];
    uint32_t *hdr = reinterpret_cast<uint32_t *>(p.buf);
    if (r == 0 && lane == 0) {
        hdr[0] = (uint32_t)total;
        hdr[1] = (uint32_t)(total >> 32);
        hdr[2] = (uint32_t)p.n;
        hdr[3] = (uint32_t)(p.path ? p.path_bytes : 0) | (p.qual ? 0x100u : 0u);
    }
    const uint64_t off = p.offsets[r];
    const int64_t len = (int64_t)(p.offsets[r + 1] - off);
    if (lane == 0) {
        hdr[4 + r] = (uint32_t)len;
        reinterpret_cast<int32_t *>(hdr + 4 + p.n)[r] = p.status ? p.status[r] : 0;
    }
    uint8_t *lab_out = p.buf + header_bytes(p.n) + off;
    const uint8_t *lab_in = p.labels + r * p.stride;
    for (int64_t j = lane; j < len; j += 64) lab_out[j] = lab_in[j];
    uint8_t *path_region = p.buf + header_bytes(p.n) + ((total + 3) & ~(uint64_t)3);
    uint64_t path_total = 0;
    if (p.path) {
        const uint32_t *pth_in = p.path + r * p.stride;
        path_total = total * (uint64_t)p.path_bytes;
        if (p.path_bytes == 2) {
            uint16_t *o = reinterpret_cast<uint16_t *>(path_region) + off;
            for (int64_t j = lane; j < len; j += 64) o[j] = (uint16_t)pth_in[j];
        } else {
            uint32_t *o = reinterpret_cast<uint32_t *>(path_region) + off;
            for (int64_t j = lane; j < len; j += 64) o[j] = pth_in[j];
        }
    }
    if (p.qual) {
        float *o = reinterpret_cast<float *>(path_region + ((path_total + 3) & ~(uint64_t)3)) + off;
        const float *q_in = p.qual + r * p.stride;
        for (int64_t j = lane; j < len; j += 64) o[j] = q_in[j];
    }
}

struct UnpackParams {
    const uint8_t *buf;
    int64_t n;
    const uint64_t *offsets;  // prefix sums of the buffer's out_len
    uint8_t *labels;
    uint32_t *path;
    uint32_t *out_len;
    int32_t *status;
    int64_t stride;
};

__global__ __launch_bounds__(256) void unpack_kernel(UnpackParams p) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= p.n) return;
    const uint32_t *hdr = reinterpret_cast<const uint32_t *>(p.buf);
    const uint64_t total = (uint64_t)hdr[0] | ((uint64_t)hdr[1] << 32);
    const int64_t n_in = hdr[2];
    const int path_bytes = (int)(hdr[3] & 0xffu);
    const uint64_t off = p.offsets[r];
    const int64_t len = (int64_t)(p.offsets[r + 1] - off);
    if (lane == 0) {
        p.out_len[r] = (uint32_t)len;
        if (p.status) p.status[r] = reinterpret_cast<const int32_t *>(hdr + 4 + n_in)[r];
    }
    const uint8_t *lab_in = p.buf + header_bytes(n_in) + off;
    uint8_t *lab_out = p.labels + r * p.stride;
    for (int64_t j = lane; j < len; j += 64) lab_out[j] = lab_in[j];
    if (!p.path || path_bytes == 0) return;
    const uint8_t *path_region = p.buf + header_bytes(n_in) + ((total + 3) & ~(uint64_t)3);
    uint32_t *pth_out = p.path + r * p.stride;
    if (path_bytes == 2) {
        const uint16_t *in = reinterpret_cast<const uint16_t *>(path_region) + off;
        for (int64_t j = lane; j < len; j += 64) pth_out[j] = in[j];
    } else {
        const uint32_t *in = reinterpret_cast<const uint32_t *>(path_region) + off;
        for (int64_t j = lane; j < len; j += 64) pth_out[j] = in[j];
    }
}

// ---- the receiving side of the gather: every shard of the gathered buffer in ONE pair of launches ----
// gathered = world buffers of `stride` bytes back to back (the layout of ncclGather / dist.gather into one
// allocation); shard s holds counts[s] reads and lands in rows [first[s], first[s] + counts[s]) of `out`.
struct GatherUnpack {
    const uint8_t *gathered;
    int64_t stride;
    int world;
    const int64_t *first;   // [world + 1] exclusive prefix sums of the per-rank read counts (device)
    uint64_t *offsets;      // workspace [n_total + world]: per shard, counts[s] + 1 prefix sums of its out_len
    uint8_t *labels;
    uint32_t *path;
    uint32_t *out_len;
    int32_t *status;
    int64_t out_stride;
    int32_t *bad;           // set to 1 + shard when a shard's header contradicts counts / the buffer size
};

// one workgroup per shard: header check + prefix sums of the shard's out_len
__global__ __launch_bounds__(kScanThreads) void gathered_offsets_kernel(GatherUnpack p) {
    __shared__ uint64_t s_part[kScanThreads / 64];
    __shared__ uint64_t s_carry;
    const int s = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n = p.first[s + 1] - p.first[s];
    const uint8_t *buf = p.gathered + (int64_t)s * p.stride;
    const uint32_t *hdr = reinterpret_cast<const uint32_t *>(buf);
    uint64_t *offs = p.offsets + p.first[s] + s;
    bool ok = true;
    uint64_t total = 0;
    if (n > 0) {
        total = (uint64_t)hdr[0] | ((uint64_t)hdr[1] << 32);
        const uint32_t pb = hdr[3] & 0xffu;
        ok = (int64_t)hdr[2] == n && (pb == 2 || pb == 4) &&
             16 + 8 * (uint64_t)n + ((total + 3) & ~(uint64_t)3) + total * pb <= (uint64_t)p.stride;
    }
    if (!ok && tid == 0) atomicMax(p.bad, 1 + s);
    if (tid == 0) s_carry = 0;
    __syncthreads();
    const uint32_t *len = hdr + 4;
    for (int64_t base = 0; base < n; base += kScanThreads) {
        const int64_t i = base + tid;
        uint64_t v = 0;
        if (i < n && ok) {
            const uint64_t l = len[i];
            v = l < (uint64_t)p.out_stride ? l : (uint64_t)p.out_stride;
        }
        uint64_t incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)incl, o);
            const uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(incl >> 32), o);
            if (lane >= o) incl += ((uint64_t)hi << 32) | lo;
        }
        if (lane == 63) s_part[wave] = incl;
        __syncthreads();
        uint64_t before = s_carry;
        for (int w = 0; w < wave; ++w) before += s_part[w];
        if (i < n) offs[i] = before + incl - v;
        __syncthreads();
        if (tid == kScanThreads - 1) s_carry = before + incl;
        __syncthreads();
    }
    // the lengths must add up to the header's total: a shard whose lengths promise more than it holds would send
    // the unpack past its end (and, for the last shard, past the gathered buffer) -- its rows are left empty instead
    __syncthreads();
    if (ok && n > 0 && s_carry != total) {
        if (tid == 0) atomicMax(p.bad, 1 + s);
        for (int64_t i = tid; i <= n; i += kScanThreads) offs[i] = 0;
    } else if (tid == 0) {
        offs[n] = s_carry;
    }
}

__global__ __launch_bounds__(256) void gathered_unpack_kernel(GatherUnpack p) {
    const int lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // global read index
    const int64_t n_total = p.first[p.world];
    if (g >= n_total) return;
    int s = 0;
    while (s + 1 < p.world && g >= p.first[s + 1]) ++s;  // world is a handful of ranks
    const int64_t r = g - p.first[s], n = p.first[s + 1] - p.first[s];
    const uint8_t *buf = p.gathered + (int64_t)s * p.stride;
    const uint32_t *hdr = reinterpret_cast<const uint32_t *>(buf);
    const uint64_t *offs = p.offsets + p.first[s] + s;
    const uint64_t total = (uint64_t)hdr[0] | ((uint64_t)hdr[1] << 32);
    const int path_bytes = (int)(hdr[3] & 0xffu);
    const uint64_t off = offs[r];
    const int64_t len = (int64_t)(offs[r + 1] - off);
    if (lane == 0) {
        p.out_len[g] = (uint32_t)len;
        if (p.status) p.status[g] = reinterpret_cast<const int32_t *>(hdr + 4 + n)[r];
    }
    const uint8_t *lab_in = buf + header_bytes(n) + off;
    uint8_t *lab_out = p.labels + g * p.out_stride;
    for (int64_t j = lane; j < len; j += 64) lab_out[j] = lab_in[j];
    if (!p.path) return;
    const uint8_t *path_region = buf + header_bytes(n) + ((total + 3) & ~(uint64_t)3);
    uint32_t *pth_out = p.path + g * p.out_stride;
    if (path_bytes == 2) {
        const uint16_t *in = reinterpret_cast<const uint16_t *>(path_region) + off;
        for (int64_t j = lane; j < len; j += 64) pth_out[j] = in[j];
    } else {
        const uint32_t *in = reinterpret_cast<const uint32_t *>(path_region) + off;
        for (int64_t j = lane; j < len; j += 64) pth_out[j] = in[j];
    }
}

}  // namespace

hipError_t launch_unpack_gathered(const uint8_t *gathered, int64_t stride, int world, const int64_t *first,
                                  int64_t n_total, uint64_t *offsets, const ResultDesc &out, int32_t *bad,
                                  hipStream_t stream) {
    if (world <= 0 || n_total <= 0) return hipSuccess;
    GatherUnpack p{gathered, stride, world, first, offsets, out.labels, out.path, out.out_len, out.status,
                   out.out_stride, bad};
    hipLaunchKernelGGL(gathered_offsets_kernel, dim3((unsigned)world), dim3(kScanThreads), 0, stream, p);
    hipLaunchKernelGGL(gathered_unpack_kernel, dim3((unsigned)((n_total + 3) / 4)), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_result_offsets(const uint32_t *len, int64_t n, int64_t stride, uint64_t *offsets, hipStream_t stream) {
    hipLaunchKernelGGL(result_offsets_kernel, dim3(1), dim3(kScanThreads), 0, stream, len, n, stride, offsets);
    return hipGetLastError();
}

hipError_t launch_pack(const ResultDesc &res, int64_t n, int path_bytes, const uint64_t *offsets, uint8_t *buf,
                       hipStream_t stream) {
    PackParams p{res.labels, res.path, res.qual, res.out_len, res.status, res.out_stride, n, path_bytes, offsets, buf};
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_unpack(const uint8_t *buf, int64_t n, const uint64_t *offsets, const ResultDesc &out,
                         hipStream_t stream) {
    UnpackParams p{buf, n, offsets, out.labels, out.path, out.out_len, out.status, out.out_stride};
    hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, p);
    return hipGetLastError();
}

}  // namespace fcd
