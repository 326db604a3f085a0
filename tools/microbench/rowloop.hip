// Micro-benchmark of the max-mode window-building row loop (duplex.hip): where do ~360 cycles per row go?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/rowloop.hip -o tools/microbench/rowloop && tools/microbench/rowloop
// Variants: bit 0 = LDS operand reads, bit 1 = global row stores, bit 2 = only 9 of 64 lanes active.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct __attribute__((packed, aligned(4))) Row3 { float a, b, c; };

template <int V>
__global__ __launch_bounds__(64) void rowloop(float *out, float *vec, uint64_t *cyc, int W, int steps, int Wcap) {
    __shared__ float w2[130 * 5];
    __shared__ float ring[8 * 130 * 3];
    const int lane = threadIdx.x;
    for (int i = lane; i < 130 * 5; i += 64) w2[i] = -0.5f - 0.001f * (float)i;
    for (int i = lane; i < 8 * 130 * 3; i += 64) ring[i] = -1.0f - 0.0001f * (float)i;
    __syncthreads();
    const bool active = (V & 4) ? (lane % 7 == 0) : true;
    float *my = vec + ((size_t)blockIdx.x * 64 + lane) * (size_t)Wcap * 3;
    const float *wq = w2;
    const float *xq = ring + (lane & 7) * 130 * 3 + 2;
    const int l = lane & 3;
    float acc = 0.0f;
    uint64_t t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) : : "memory");
    if (active) {
        for (int s = 0; s < steps; ++s) {
            float l_lab = -__builtin_huge_valf(), l_sum = -__builtin_huge_valf(), mx = -__builtin_huge_valf();
            int s3 = 3 * (s % Wcap);
            for (int j = 0; j < W; ++j) {
                float c0 = -0.7f, cl = -0.9f, cx = -1.1f;
                if (V & 1) {
                    c0 = wq[j * 5];
                    cl = wq[j * 5 + l + 1];
                    cx = xq[s3];
                }
                const float g = l_sum + c0;
                float m1, sm;
                asm("v_max_f32 %0, %1, %2" : "=v"(m1) : "v"(l_lab), "v"(cx));
                const float lb = cl + m1;
                asm("v_max_f32 %0, %1, %2" : "=v"(sm) : "v"(lb), "v"(g));
                if (V & 2) *reinterpret_cast<Row3 *>(my + s3) = Row3{lb, g, sm};
                asm("v_max_f32 %0, %1, %2" : "=v"(mx) : "v"(mx), "v"(sm));
                l_lab = lb;
                l_sum = sm;
                s3 = s3 + 3 == 3 * Wcap ? 0 : s3 + 3;
            }
            acc += mx;
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : : "memory");
    out[blockIdx.x * 64 + lane] = acc;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V>
int run(int blocks, int W, int steps) {
    const int Wcap = 130;
    float *out, *vec;
    uint64_t *cyc;
    CHECK(hipMalloc(&out, blocks * 64 * 4));
    CHECK(hipMalloc(&vec, (size_t)blocks * 64 * Wcap * 12));
    CHECK(hipMalloc(&cyc, blocks * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int it = 0; it < 2; ++it) {
        CHECK(hipEventRecord(e0));
        rowloop<V><<<blocks, 64>>>(out, vec, cyc, W, steps, Wcap);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
    }
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint64_t> h(blocks);
    CHECK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
    double m = 0;
    for (auto v : h) m += (double)v;
    m /= blocks;
    printf("blocks %d variant lds=%d store=%d few_lanes=%d: %.3f ms, %.1f counter ticks per row, %.1f ns per row\n", blocks, V & 1, (V >> 1) & 1,
           (V >> 2) & 1, ms, m / ((double)W * steps), ms * 1e6 / ((double)W * steps));
    hipFree(out); hipFree(vec); hipFree(cyc);
    return 0;
}

int main() {
    const int W = 128, steps = 200;
    for (int blocks : {256, 1024}) {
        if (run<0>(blocks, W, steps)) return 1;
        if (run<1>(blocks, W, steps)) return 1;
        if (run<2>(blocks, W, steps)) return 1;
        if (run<3>(blocks, W, steps)) return 1;
        if (run<7>(blocks, W, steps)) return 1;
        if (run<4>(blocks, W, steps)) return 1;
    }
    return 0;
}
