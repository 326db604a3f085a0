// tieorder.hip -- test hook for csrc/pdq178.h (the tie order of Rust 1.78's sort_unstable_by): sorts lists with
// the very device function the beam kernels call on their tie-flagged steps, the list living in LDS as it does
// there (lists too long for LDS are sorted in place in HBM).  fcd_debug_pdq178_sort_dev, include/fcd.h.
#include "device_utils.h"
#include "fcd_internal.h"
#include "pdq178.h"
#define FCD_WAVE_PROF 1  // (this translation unit only: the probe kernels; the search kernels carry no stamps)
#include "pdq178_wave.h"

namespace fcd {

namespace {

constexpr int kProbeLds = 2048;  // elements

__global__ __launch_bounds__(64) void pdq178_probe_kernel(uint64_t *lists, int64_t stride, const int32_t *lens) {
    __shared__ uint64_t s_list[kProbeLds];
    __shared__ pdq178::Scratch s_scr;
    uint64_t *list = lists + (int64_t)blockIdx.x * stride;
    const int n = lens[blockIdx.x];
    const int lane = threadIdx.x;
    if (n <= kProbeLds) {
        for (int j = lane; j < n; j += kWave) s_list[j] = list[j];
        __syncthreads();
        if (lane == 0) pdq178::sort_desc(s_list, n, &s_scr);
        __syncthreads();
        for (int j = lane; j < n; j += kWave) list[j] = s_list[j];
    } else if (lane == 0) {
        pdq178::sort_desc(list, n, &s_scr);
    }
}

// the wave-cooperative routine (pdq178_wave.h): block b sorts list b, living in LDS at an offset that is not a
// multiple of 64 (like the second read's list of a wavefront in the wide-beam kernel)
template <int P>
__global__ __launch_bounds__(64) void pdq178_wave_probe_kernel(uint64_t *lists, int64_t n_lists, int64_t stride,
                                                               const int32_t *lens, int keep) {
    __shared__ uint64_t s_v[64 * P + 40];
    __shared__ pdq178::WaveScratch<P> s_scr;
    const int lane = threadIdx.x;
    const int64_t i0 = blockIdx.x;
    const int len0 = lens[i0];
    uint64_t *v = s_v + (blockIdx.x % 3) * 20;
    for (int j = lane; j < len0; j += kWave) v[j] = lists[i0 * stride + j];
    __syncthreads();
    pdq178::wave_sort<P>(v, len0, keep, &s_scr, lane);
    __syncthreads();
    for (int j = lane; j < len0; j += kWave) lists[i0 * stride + j] = v[j];
}

}  // namespace

hipError_t launch_pdq178_coop_probe(uint64_t *lists, int64_t n_lists, int64_t stride, const int32_t *lens, int planes,
                                    int keep, hipStream_t stream) {
    if (n_lists <= 0) return hipSuccess;
    const dim3 grid((unsigned)n_lists), block(64);
    switch (planes) {
        case 1: hipLaunchKernelGGL(pdq178_wave_probe_kernel<1>, grid, block, 0, stream, lists, n_lists, stride, lens, keep); break;
        case 3: hipLaunchKernelGGL(pdq178_wave_probe_kernel<3>, grid, block, 0, stream, lists, n_lists, stride, lens, keep); break;
        case 5: hipLaunchKernelGGL(pdq178_wave_probe_kernel<5>, grid, block, 0, stream, lists, n_lists, stride, lens, keep); break;
        case 8: hipLaunchKernelGGL(pdq178_wave_probe_kernel<8>, grid, block, 0, stream, lists, n_lists, stride, lens, keep); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t coop_prof_read(unsigned long long *out16, bool reset) {
#ifndef FCD_HIPEMU
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(pdq178::g_wave_prof), 16 * sizeof(unsigned long long));
    if (e != hipSuccess) return e;
    if (reset) {
        unsigned long long zero[16] = {0};
        return hipMemcpyToSymbol(HIP_SYMBOL(pdq178::g_wave_prof), zero, sizeof(zero));
    }
    return hipSuccess;
#else
    for (int i = 0; i < 16; ++i) out16[i] = 0;
    return hipSuccess;
#endif
}

hipError_t launch_pdq178_probe(uint64_t *lists, int64_t n_lists, int64_t stride, const int32_t *lens,
                               hipStream_t stream) {
    if (n_lists <= 0) return hipSuccess;
    hipLaunchKernelGGL(pdq178_probe_kernel, dim3((unsigned)n_lists), dim3(64), 0, stream, lists, stride, lens);
    return hipGetLastError();
}

// this translation unit's copy of the replay's std-form word (pdq178.h), on the current device
FCD_PDQ178_DEFINE_STD_FORM_SETTER(tieorder_set_pdq178_std_form)

}  // namespace fcd
