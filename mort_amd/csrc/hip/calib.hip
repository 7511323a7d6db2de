/*
 * calib.hip -- two calibration kernels for the roofline model that bench.py prints beside the render kernels' counters
 * (measurement only: nothing on the render path calls them).
 *
 *   mort_hip_calib_valu      how many shader cycles one SIMD needs per wave64 VALU instruction, measured in-kernel
 *                            (s_memtime) with exactly 1..8 waves resident per SIMD and four instruction mixes.  The
 *                            guide's figure is 2 cycles (SIMD-32, MI355X_MICROARCH.md "Wave scheduling") once enough
 *                            waves share the SIMD and 4 for a wave that has it to itself; round 2's model assumed 4
 *                            throughout and printed a saturated VALU for a kernel that was at half of it.
 *   mort_hip_calib_hbm_copy  a float4 copy over buffers far larger than the 256 MB Infinity Cache: the HBM rate this
 *                            box reaches (SURVEY 8d asks for a measured peak beside the 8 TB/s specification).
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <map>
#include <utility>
#include <vector>

#include "mort_hip.h"
#include "mort_ctx.h"

#define CALIB_UNROLL 8
#define CALIB_CHAINS 16

/* one resident block per (CU, slot): 256 threads = one wave per SIMD, and the dynamic LDS request admits exactly
 * `waves_per_simd` blocks per CU */
template <int KIND>
__global__ void __launch_bounds__(256) calib_valu_kernel(int iters, float x, float y, unsigned long long *out) {
    extern __shared__ unsigned char calib_lds[];
    float a[CALIB_CHAINS];
    double d[CALIB_CHAINS / 2];
#pragma unroll
    for (int k = 0; k < CALIB_CHAINS; k++) a[k] = (float)(threadIdx.x + k);
#pragma unroll
    for (int k = 0; k < CALIB_CHAINS / 2; k++) d[k] = (double)(threadIdx.x + k);
    const double xd = (double)x, yd = (double)y;
    int sacc = iters;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < CALIB_UNROLL; u++) {
            if (KIND == 0) { /* 16 independent v_fma_f32 */
#pragma unroll
                for (int k = 0; k < CALIB_CHAINS; k++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(x), "v"(y));
            } else if (KIND == 1) { /* one dependent chain of 16 v_fma_f32 */
#pragma unroll
                for (int k = 0; k < CALIB_CHAINS; k++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(x), "v"(y));
            } else if (KIND == 2) { /* 8 independent v_fma_f64, twice */
#pragma unroll
                for (int k = 0; k < CALIB_CHAINS; k++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[k & 7]) : "v"(xd), "v"(yd));
            } else if (KIND == 4) { /* 8 independent v_pk_fma_f32 (two fp32 fma per lane and instruction), twice */
#pragma unroll
                for (int k = 0; k < CALIB_CHAINS; k++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d[k & 7]) : "v"(xd), "v"(yd));
            } else { /* 12 independent v_fma_f32 with 4 scalar instructions between them (the state machine's mix: one SALU per three VALU) */
#pragma unroll
                for (int k = 0; k < CALIB_CHAINS; k++) {
                    if ((k & 3) == 3) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc) : : "scc");
                    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(x), "v"(y));
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
#pragma unroll
    for (int k = 0; k < CALIB_CHAINS; k++) s += a[k];
#pragma unroll
    for (int k = 0; k < CALIB_CHAINS / 2; k++) s += (float)d[k];
    if (s == 12345.678f && sacc == 7) out[0] = 1; /* keeps the chains alive */
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        /* where the wave ran: HW_REG_HW_ID (id 4: simd [5:4], cu [11:8], sh [12], se [15:13]) and HW_REG_XCC_ID (id 20, [3:0]) */
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
        out[8 + 4 * w] = t1 - t0;
        out[9 + 4 * w] = r0;
        out[10 + 4 * w] = r1;
        out[11 + 4 * w] = ((unsigned long long)(xcc & 0xfu) << 16) | (unsigned long long)(hw & 0xff30u);
    }
}

extern "C" int mort_hip_calib_valu(mort_ctx *c, int waves_per_simd, int kind, mort_calib_valu *res) {
    if (!c || !res || waves_per_simd < 1 || waves_per_simd > 8 || kind < 0 || kind > 4) return MORT_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    void (*kern)(int, float, float, unsigned long long *) =
        kind == 0 ? calib_valu_kernel<0> : kind == 1 ? calib_valu_kernel<1> : kind == 2 ? calib_valu_kernel<2> : kind == 4 ? calib_valu_kernel<4> : calib_valu_kernel<3>;
    const int blocks = c->num_cus * waves_per_simd;
    /* exactly waves_per_simd blocks fit a CU's 160 KB (the hardware hands LDS out in granules: a request that is not a multiple of
     * 2 KB is rounded up, and one block fewer fits than the byte count says) */
    const size_t lds = ((size_t)160 * 1024 / (size_t)waves_per_simd) & ~(size_t)2047;
    HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0;
    HIPCHK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, lds));
    if (per_cu != waves_per_simd) { c->last_error = "calib_valu: occupancy query disagrees with the LDS sizing"; return MORT_ERR_HIP; }
    unsigned long long *d_out = nullptr;
    const size_t n_out = 8 + 4 * (size_t)blocks * 4;
    HIPCHK(c, hipMalloc((void **)&d_out, n_out * sizeof(unsigned long long)));
    const int iters = 20000 / waves_per_simd + 2000;
    int st = MORT_OK;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    std::vector<unsigned long long> h(n_out);
    float ms = 0;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) st = MORT_ERR_HIP;
    for (int rep = 0; rep < 2 && st == MORT_OK; rep++) { /* the second launch is the measured one */
        hipMemsetAsync(d_out, 0, n_out * sizeof(unsigned long long), c->stream);
        hipEventRecord(e0, c->stream);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, c->stream, iters, 1.0000001f, 1e-9f, d_out);
        hipEventRecord(e1, c->stream);
        if (hipGetLastError() != hipSuccess || hipEventSynchronize(e1) != hipSuccess) st = MORT_ERR_HIP;
    }
    if (st == MORT_OK && (hipEventElapsedTime(&ms, e0, e1) != hipSuccess ||
                          hipMemcpy(h.data(), d_out, n_out * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess)) st = MORT_ERR_HIP;
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    hipFree(d_out);
    if (st != MORT_OK) return st;
    /* per wave: loop cycles and its clock; per SIMD (xcc, se, sh, cu, simd of HW_ID): the waves that ran there, how many of them at
     * once, and the SIMD's cycles per instruction = (its first start .. its last end) x clock / instructions issued there */
    std::vector<double> cyc, clk;
    struct Simd { double first = 1e300, last = 0; int waves = 0; std::vector<std::pair<double, int>> ev; };
    std::map<unsigned long long, Simd> simds;
    for (size_t w = 0; w < (size_t)blocks * 4; w++) {
        const double t = (double)h[8 + 4 * w], r0 = (double)h[9 + 4 * w], r1 = (double)h[10 + 4 * w];
        if (!(t > 0 && r1 > r0)) continue;
        cyc.push_back(t); clk.push_back(t / (r1 - r0) * 0.1); /* s_memrealtime ticks at 100 MHz */
        Simd &sd = simds[h[11 + 4 * w]];
        sd.first = std::min(sd.first, r0); sd.last = std::max(sd.last, r1); sd.waves++;
        sd.ev.push_back({r0, +1}); sd.ev.push_back({r1, -1});
    }
    if (cyc.empty()) return MORT_ERR_HIP;
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double valu_per_wave = (double)iters * CALIB_UNROLL * (kind == 3 ? 12.0 : (double)CALIB_CHAINS);
    const double clock = clk[clk.size() / 2];
    std::vector<double> cpi, conc;
    for (auto &kv : simds) {
        Simd &sd = kv.second;
        cpi.push_back((sd.last - sd.first) * 10.0 * clock / (valu_per_wave * sd.waves)); /* ticks x 10 ns x GHz = cycles */
        std::sort(sd.ev.begin(), sd.ev.end());
        int cur = 0, mx = 0;
        for (auto &e : sd.ev) { cur += e.second; mx = std::max(mx, cur); }
        conc.push_back((double)mx);
    }
    std::sort(cpi.begin(), cpi.end()); std::sort(conc.begin(), conc.end());
    res->waves_per_simd = waves_per_simd; res->kind = kind;
    res->seconds = ms * 1e-3;
    res->cycles_per_wave = cyc[cyc.size() / 2];
    res->clock_ghz = clock;
    res->valu_per_wave = valu_per_wave;
    res->simds_seen = (int)simds.size();
    res->resident_waves_per_simd = conc[conc.size() / 2];
    res->cycles_per_valu_per_wave = res->cycles_per_wave / valu_per_wave;
    res->cycles_per_valu_per_simd = cpi[cpi.size() / 2];
    return MORT_OK;
}

__global__ void __launch_bounds__(256) calib_copy_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n16) {
    /* four 16-byte loads in flight per lane before the stores */
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

extern "C" int mort_hip_calib_hbm_copy(mort_ctx *c, size_t bytes, int reps, double *gbs_out) {
    if (!c || !gbs_out || bytes < (1u << 20) || reps < 1) return MORT_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    void *a = nullptr, *b = nullptr;
    HIPCHK(c, hipMalloc(&a, bytes));
    if (hipMalloc(&b, bytes) != hipSuccess) { hipFree(a); return MORT_ERR_NOMEM; }
    int st = MORT_OK;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipMemsetAsync(a, 1, bytes, c->stream) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) st = MORT_ERR_HIP;
    const size_t n16 = bytes / 16;
    const int grid = c->num_cus * 16;
    double best = 0;
    for (int r = 0; r <= reps && st == MORT_OK; r++) { /* r = 0 warms up */
        hipEventRecord(e0, c->stream);
        hipLaunchKernelGGL(calib_copy_kernel, dim3(grid), dim3(256), 0, c->stream, (const float4 *)a, (float4 *)b, n16);
        hipEventRecord(e1, c->stream);
        float ms = 0;
        if (hipGetLastError() != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { st = MORT_ERR_HIP; break; }
        if (r > 0 && ms > 0) best = std::max(best, 2.0 * (double)(n16 * 16) / (ms * 1e-3) / 1e9); /* read + write */
    }
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    hipFree(a); hipFree(b);
    if (st == MORT_OK) *gbs_out = best;
    return st;
}
