// What does a pure streaming read reach on this GPU?  (context for the viterbi kernel's HBM fraction)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/readbw.hip -o tools/microbench/readbw && tools/microbench/readbw
// Reads 1.31 GB (16384 reads x 4000 x 5 floats, the viterbi bench shape) with 16-byte loads and folds them into one
// value per thread: (a) flat grid-stride over the whole buffer, (b) one workgroup per 80 KB read (the viterbi kernel's
// mapping), at several workgroup sizes / loads in flight.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int UNROLL>
__global__ void flat(const uint4 *p, size_t n, uint32_t *out) {
    uint32_t acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p + i + u * stride));
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n; i += stride) {
        const uint4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;  // (never: keeps the loads alive)
}

template <int UNROLL>
__global__ void per_read(const uint4 *p, size_t per, uint32_t *out) {
    const uint4 *q = p + (size_t)blockIdx.x * per;
    uint32_t acc = 0;
    size_t i = threadIdx.x;
    for (; i + (UNROLL - 1) * blockDim.x < per; i += UNROLL * blockDim.x) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(q + i + u * blockDim.x));
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < per; i += blockDim.x) {
        const uint4 v = q[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const size_t reads = 16384, per = 4000 * 5 * 4 / 16;  // 16-byte units per read
    const size_t n = reads * per;
    uint4 *buf;
    uint32_t *out;
    CHECK(hipMalloc(&buf, n * 16));
    CHECK(hipMalloc(&out, 4));
    CHECK(hipMemset(buf, 1, n * 16));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto time = [&](auto launch, const char *name) -> int {
        float best = 1e9f;
        for (int it = 0; it < 6; ++it) {
            CHECK(hipEventRecord(e0));
            launch();
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (it > 0 && ms < best) best = ms;
        }
        printf("%-44s %.4f ms  %.2f TB/s\n", name, best, n * 16 / best / 1e9);
        return 0;
    };
    time([&] { flat<4><<<256 * 8, 256>>>(buf, n, out); }, "flat grid-stride, 2048 x 256, 4 loads");
    time([&] { flat<8><<<256 * 8, 256>>>(buf, n, out); }, "flat grid-stride, 2048 x 256, 8 loads");
    time([&] { flat<4><<<256 * 16, 256>>>(buf, n, out); }, "flat grid-stride, 4096 x 256, 4 loads");
    time([&] { flat<8><<<256 * 4, 512>>>(buf, n, out); }, "flat grid-stride, 1024 x 512, 8 loads");
    time([&] { flat<4><<<256 * 32, 64>>>(buf, n, out); }, "flat grid-stride, 8192 x 64, 4 loads");
    time([&] { per_read<4><<<reads, 64>>>(buf, per, out); }, "one 64-thread workgroup per read, 4 loads");
    time([&] { per_read<8><<<reads, 64>>>(buf, per, out); }, "one 64-thread workgroup per read, 8 loads");
    time([&] { per_read<4><<<reads, 128>>>(buf, per, out); }, "one 128-thread workgroup per read, 4 loads");
    time([&] { per_read<4><<<reads, 256>>>(buf, per, out); }, "one 256-thread workgroup per read, 4 loads");
    return 0;
}
