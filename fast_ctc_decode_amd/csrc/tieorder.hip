// tieorder.hip -- test hook for csrc/pdq178.h (the tie order of Rust 1.78's sort_unstable_by): sorts lists with
// the very device function the beam kernels call on their tie-flagged steps, the list living in LDS as it does
// there (lists too long for LDS are sorted in place in HBM).  fcd_debug_pdq178_sort_dev, include/fcd.h.
#include "device_utils.h"
#include "fcd_internal.h"
#include "pdq178.h"

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

}  // namespace

hipError_t launch_pdq178_probe(uint64_t *lists, int64_t n_lists, int64_t stride, const int32_t *lens,
                               hipStream_t stream) {
    if (n_lists <= 0) return hipSuccess;
    hipLaunchKernelGGL(pdq178_probe_kernel, dim3((unsigned)n_lists), dim3(64), 0, stream, lists, stride, lens);
    return hipGetLastError();
}

}  // namespace fcd
