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

/* The "heavy" head of the cost order for the unified-tree megakernel's heavy waves (mega_gen.hip, FastArgs.heavy_*): how many tiles of the
 * sorted (descending) list of per-tile longest-pixel costs reach `percent` % of the largest one, at most max_r -- or 0 when the frame is not
 * bound by its longest pixel chains: the longest pixel must be at least twice an average lane's share of the frame that produced these costs
 * (*frame_total segments over `lanes` lanes).  One workgroup. */
__global__ void __launch_bounds__(256) heavy_count_kernel(const unsigned *keys_desc, int n, unsigned percent, unsigned max_r,
                                                          const unsigned long long *frame_total, unsigned long long lanes, unsigned *out) {
    __shared__ unsigned s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const unsigned long long top = keys_desc[0];
    const unsigned long long thr = (top * percent + 99ull) / 100ull;
    unsigned c = 0;
    for (int i = threadIdx.x; i < n; i += 256) c += ((unsigned long long)keys_desc[i] >= thr && keys_desc[i] > 0u) ? 1u : 0u;
    atomicAdd(&s_cnt, c);
    __syncthreads();
    if (threadIdx.x == 0) {
        /* frame_total: a saved copy of the launch's 96 counters -- segments = [0] + the 32 per-workgroup slots [32 + 2 k] (mort_hip.hip) */
        unsigned long long total = 0ull;
        if (frame_total) { total = frame_total[0]; for (int k = 0; k < 32; k++) total += frame_total[32 + 2 * k]; }
        const bool chain_bound = total > 0ull && top * lanes >= 2ull * total;
        *out = chain_bound ? (s_cnt < max_r ? s_cnt : max_r) : 0u;
    }
}
hipError_t mort_tile_heavy_count(const unsigned *d_keys_desc, int n, unsigned percent, unsigned max_r, const unsigned long long *d_frame_total,
                                 unsigned long long lanes, unsigned *d_out, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(heavy_count_kernel, dim3(1), dim3(256), 0, s, d_keys_desc, n, percent, max_r, d_frame_total, lanes, d_out);
    return hipGetLastError();
}
