// tools/microbench/branchlat.hip -- one wavefront alone on its SIMD: what a scalar instruction, a taken branch, a branch
// that falls through, a v_readlane -> scalar use and an LDS cross-lane round trip cost.  The quicksort replay
// (csrc/pdq178_reg.h) is ~650 mostly scalar instructions per partition with dozens of short branches: this says whether its
// cycles are its instructions or its branches.   hipcc --offload-arch=gfx950 -O3 branchlat.hip -o branchlat
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP16(x) x x x x x x x x x x x x x x x x
#define T0 asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) : : "memory")
#define T1 asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : : "memory")
__global__ void k(uint64_t *cyc, int *sink, int a) {
    uint64_t t0, t1;
    int s = a, v = threadIdx.x + a;
    // 0: dependent s_add
    T0;
    for (int i = 0; i < 64; ++i) asm volatile(REP16("s_add_u32 %0, %0, 1\n\t") : "+s"(s));
    T1; if (threadIdx.x == 0) cyc[0] = t1 - t0;
    // 1: taken forward branch to the next instruction
    T0;
    for (int i = 0; i < 64; ++i) asm volatile(REP16("s_branch 0\n\t") ::: "memory");
    T1; if (threadIdx.x == 0) cyc[1] = t1 - t0;
    // 2: conditional branch that falls through (scc = 0)
    T0;
    for (int i = 0; i < 64; ++i) asm volatile("s_cmp_eq_u32 0, 1\n\t" REP16("s_cbranch_scc1 0\n\t") ::: "memory", "scc");
    T1; if (threadIdx.x == 0) cyc[2] = t1 - t0;
    // 3: conditional branch taken over one skipped instruction
    T0;
    for (int i = 0; i < 64; ++i) asm volatile("s_cmp_eq_u32 0, 0\n\t" REP16("s_cbranch_scc1 1\n\ts_nop 0\n\t") ::: "memory", "scc");
    T1; if (threadIdx.x == 0) cyc[3] = t1 - t0;
    // 4: v_readlane -> s_add on the result -> v_add with the scalar (the replay's sample / position traffic)
    T0;
    for (int i = 0; i < 64; ++i)
        asm volatile(REP16("v_readlane_b32 %1, %0, 3\n\ts_add_u32 %1, %1, 1\n\tv_add_u32 %0, %1, %0\n\t") : "+v"(v), "+s"(s));
    T1; if (threadIdx.x == 0) cyc[4] = t1 - t0;
    // 5: dependent ds_bpermute round trips
    T0;
    for (int i = 0; i < 64; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) v = __builtin_amdgcn_ds_bpermute((v & 63) << 2, v + 1);
    }
    T1; if (threadIdx.x == 0) cyc[5] = t1 - t0;
    // 6: dependent v_add (VALU)
    T0;
    for (int i = 0; i < 64; ++i) asm volatile(REP16("v_add_u32 %0, 1, %0\n\t") : "+v"(v));
    T1; if (threadIdx.x == 0) cyc[6] = t1 - t0;
    // 7: exec-mask branch as the compiler emits for `if (lane == k) {...}`: s_and_saveexec + s_cbranch_execz (taken: all lanes off)
    T0;
    for (int i = 0; i < 64; ++i)
        asm volatile(REP16("s_mov_b64 s[20:21], exec\n\ts_mov_b64 exec, 0\n\ts_cbranch_execz 1\n\tv_add_u32 %0, 1, %0\n\ts_mov_b64 exec, s[20:21]\n\t") : "+v"(v) :: "s20", "s21");
    T1; if (threadIdx.x == 0) cyc[7] = t1 - t0;
    // 8: 64-bit scalar mask work: s_lshl_b64 + s_bcnt1 + s_ff1 (dependent)
    uint64_t m = 0x123456789abcdefull + a;
    T0;
    for (int i = 0; i < 64; ++i)
        asm volatile(REP16("s_lshl_b64 %0, %0, 1\n\ts_bcnt1_i32_b64 %1, %0\n\ts_ff1_i32_b64 %1, %0\n\t") : "+s"(m), "+s"(s) :: "scc");
    T1; if (threadIdx.x == 0) cyc[8] = t1 - t0;
    sink[threadIdx.x] = s + v + (int)m;
}
int main() {
    uint64_t *cyc, h[9]; int *sink;
    hipMalloc(&cyc, 72); hipMalloc(&sink, 256);
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, cyc, sink, 1); hipDeviceSynchronize(); }
    hipMemcpy(h, cyc, 72, hipMemcpyDeviceToHost);
    const char *name[9] = {"s_add_u32 (dependent)", "s_branch, taken, to the next instruction", "s_cbranch_scc1, not taken", "s_cbranch_scc1 taken over one s_nop (pair)",
                           "v_readlane -> s_add -> v_add (triple)", "ds_bpermute round trip (dependent)", "v_add_u32 (dependent)",
                           "exec = 0; s_cbranch_execz taken; restore (5 instructions)", "s_lshl_b64 + s_bcnt1 + s_ff1 (triple)"};
    for (int i = 0; i < 9; ++i) printf("%-62s %.2f cycles each\n", name[i], (double)h[i] / 1024.0);
    return 0;
}
