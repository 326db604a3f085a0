// What bounds the viterbi streaming kernel?  Its memory shape in isolation (one wavefront per 80 KB read, 256-row
// tiles of N = 5 floats prefetched one or two tiles ahead), with the pieces switched on one by one:
//   bit 0: park the tile in LDS and read it back row-per-lane      bit 1: the output stores (~32 labels + path words
//   per 64 rows)      bit 2: two tiles in flight instead of one
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/vitshape.hip -o tools/microbench/vitshape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int WPB = 4;

template <int V>
__global__ __launch_bounds__(64 * WPB) void shape(const char *in, uint8_t *lab, uint32_t *pth, int T, uint32_t *sink) {
    __shared__ __attribute__((aligned(16))) float s_tile[WPB][1280];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t r = (int64_t)blockIdx.x * WPB + wave;
    const char *post = in + r * (int64_t)T * 20;
    uint8_t *lb = lab + r * (int64_t)T;
    uint32_t *pt = pth + r * (int64_t)T;
    float *tile = s_tile[wave];
    const int tiles = T / 256;
    uint4 a[5], b[5], c[5];
    auto fetch = [&](int t, uint4 (&v)[5]) {
        const char *src = post + (int64_t)t * 5120 + lane * 16;
#pragma unroll
        for (int m = 0; m < 5; ++m) v[m] = *reinterpret_cast<const uint4 *>(src + m * 1024);
    };
    uint32_t acc = 0;
    int n_out = 0;
    fetch(0, a);
    if (V & 4) { if (tiles > 1) fetch(1, b); }
    for (int t = 0; t < tiles; ++t) {
        if (V & 4) { if (t + 2 < tiles) fetch(t + 2, c); }
        else { if (t + 1 < tiles) fetch(t + 1, b); }
        if (V & 1) {
#pragma unroll
            for (int m = 0; m < 5; ++m) *reinterpret_cast<uint4 *>(tile + (lane + 64 * m) * 4) = a[m];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float best;
            int label = 0;
            if (V & 1) {
                const float *pr = tile + (64 * u + lane) * 5;
                best = pr[0];
#pragma unroll
                for (int j = 1; j < 5; ++j) { const float v = pr[j]; if (v > best) { best = v; label = j; } }
            } else {
                best = __uint_as_float(a[u].x ^ a[u].y ^ a[u].z ^ a[u].w ^ a[4].x);
                label = (int)(a[u].x >> 3) & 3;
            }
            acc += __float_as_uint(best);
            if (V & 2) {
                const bool emit = ((lane + u + label) & 1) != 0;  // about half the rows emit
                const uint64_t m = __ballot(emit);
                const uint32_t my = (uint32_t)n_out + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
                if (emit) { lb[my] = (uint8_t)label; pt[my] = (uint32_t)(t * 256 + 64 * u + lane); }
                n_out += __builtin_popcountll(m);
            }
        }
        if (V & 8) {  // the same output bytes as 128 labels + 128 path words per tile, written as whole aligned lines
            reinterpret_cast<uint16_t *>(lb + (size_t)t * 128)[lane] = (uint16_t)acc;
            reinterpret_cast<uint2 *>(pt + (size_t)t * 128)[lane] = make_uint2(acc, (uint32_t)t);
        }
        if (V & 1) __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int m = 0; m < 5; ++m) { a[m] = b[m]; if (V & 4) b[m] = c[m]; }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    const int reads = 16384, T = 4096;  // (whole tiles only)
    char *in; uint8_t *lab; uint32_t *pth, *sink;
    const size_t nin = (size_t)reads * T * 20;
    CHECK(hipMalloc(&in, nin));
    CHECK(hipMalloc(&lab, (size_t)reads * T));
    CHECK(hipMalloc(&pth, (size_t)reads * T * 4));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(in, 1, nin));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto time = [&](auto launch, const char *name, double out_bytes) -> int {
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
        printf("%-52s %.4f ms  read %.2f TB/s  read+write %.2f TB/s\n", name, best, nin / best / 1e9, (nin + out_bytes) / best / 1e9);
        return 0;
    };
    const double ob = (double)reads * T * 0.5 * 5;
    const dim3 g(reads / WPB), b(64 * WPB);
    time([&] { shape<0><<<g, b>>>(in, lab, pth, T, sink); }, "loads only, one tile ahead", 0);
    time([&] { shape<4><<<g, b>>>(in, lab, pth, T, sink); }, "loads only, two tiles ahead", 0);
    time([&] { shape<2><<<g, b>>>(in, lab, pth, T, sink); }, "loads + output stores", ob);
    time([&] { shape<8><<<g, b>>>(in, lab, pth, T, sink); }, "loads + the same bytes as whole aligned lines per tile", ob);
    return 0;
}
