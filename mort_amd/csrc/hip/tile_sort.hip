/*
 * tile_sort.hip -- device-side argsort of the per-tile costs (segments traced per 8x8 pixel tile in the previous
 * frame): order = tiles by decreasing cost, equal costs in index order.  The megakernels hand out tiles in that
 * order so a frame does not end on its longest pixel chains (DESIGN.md "Cost-ordered tiles").  Runs on the render
 * stream, so mort_hip_render_device stays asynchronous (round 1 sorted on the host behind a stream sync).
 * rocPRIM's radix sort is stable, which gives the index-order tie rule.
 */
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "mort_internal.h"

__global__ void __launch_bounds__(256) iota_kernel(unsigned *v, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) v[i] = (unsigned)i;
}

size_t mort_tile_sort_temp_bytes(int n) {
    size_t bytes = 0;
    unsigned *nul = nullptr;
    if (rocprim::radix_sort_pairs_desc(nullptr, bytes, nul, nul, nul, nul, (size_t)n, 0, 32, nullptr) != hipSuccess) return 0;
    return bytes;
}

hipError_t mort_tile_sort_desc(const unsigned *d_cost, unsigned *d_keys_out, unsigned *d_iota, unsigned *d_order, void *d_temp,
                               size_t temp_bytes, int n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(iota_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_iota, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return rocprim::radix_sort_pairs_desc(d_temp, temp_bytes, d_cost, d_keys_out, d_iota, d_order, (size_t)n, 0, 32, s);
}
