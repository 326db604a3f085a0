// tools/microbench/f64lat.hip -- one wavefront alone on its SIMD: cycles per v_fma_f64 when every operation waits for the
// previous one (1 chain) and when 2 / 4 independent chains alternate; the same for v_fma_f32 and v_max_f32.
// What a lone wavefront pays per instruction decides how the duplex kernels' log-add is laid out (Estrin: fewer levels,
// more operations; Horner: fewer operations, every one dependent).   hipcc --offload-arch=gfx950 -O3 f64lat.hip -o f64lat
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int CH, class T>
__global__ void k(T *out, uint64_t *cyc, T a, T b) {
    T x[CH];
    for (int c = 0; c < CH; ++c) x[c] = a + (T)c + (T)threadIdx.x;
    uint64_t t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) : : "memory");
#pragma unroll 1
    for (int i = 0; i < 64; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) x[c] = __builtin_fma(x[c], b, a);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : : "memory");
    T s = 0;
    for (int c = 0; c < CH; ++c) s += x[c];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int CH, class T>
void run(const char *name) {
    T *out; uint64_t *cyc, h;
    hipMalloc(&out, 64 * sizeof(T)); hipMalloc(&cyc, 8);
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL((k<CH, T>), dim3(1), dim3(64), 0, 0, out, cyc, (T)0.5, (T)0.999); hipDeviceSynchronize(); }
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s, %d chain(s): %.2f cycles per instruction (%d instructions)\n", name, CH, (double)h / (64.0 * 16 * CH), 64 * 16 * CH);
}
int main() {
    run<1, double>("v_fma_f64"); run<2, double>("v_fma_f64"); run<4, double>("v_fma_f64");
    run<1, float>("v_fma_f32"); run<2, float>("v_fma_f32"); run<4, float>("v_fma_f32");
    return 0;
}
