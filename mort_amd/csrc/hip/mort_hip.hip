/*
 * mort_hip.hip -- libmort_hip.so: the C ABI of include/mort_hip.h and the
 * gfx950 kernels behind it.
 *
 * Kernels (hand-written for CDNA4, wave64):
 *   seed_kernel   per-pixel XORWOW init with the 2^67-step sequence skip
 *                 (replaces setup_rng, rng.cuh:8-15)
 *   mega_kernel   one lane per pixel, 8x8 pixel tile per wave; the whole
 *                 sample x bounce nest of Camera::render / ray_color
 *                 (camera.cuh:86-208) runs as ONE flat per-lane loop so that a
 *                 lane whose path ends starts its next sample immediately
 *                 instead of idling until the longest path of the wave ends.
 *                 RNG state in 6 VGPRs for the pixel's lifetime; bounce stack
 *                 private; no global scratch arrays (mort.cu:712-725).
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "mort_hip.h"
#include "dev_pixel.h"
#include "scene_blob.h"
#include "seed_host.h"
#include "mort_internal.h"
#include "mort_ctx.h"
#include "mega_gen.h"
#include "wave_gen.h"

#pragma clang fp contract(off)

/* ====================================================================== device */

/* 4 waves per SIMD: 128 registers and 9 bounce-stack levels in LDS (36 KB per workgroup, four workgroups per CU).  Without a bound the
 * compiler takes 172 registers (two waves); at three (148 registers, 12 levels) the Cornell box 800x800x1000 took 360 ms, at four 326 ms,
 * Cornell smoke 52.9 -> 47.2 ms -- a wave issues a dependent vector instruction only every ~8 cycles, so the fourth
 * wave outweighs the extra spills (this kernel keeps its deep stack levels in private memory either way) */
#ifndef MORT_GENERIC_WAVES
#define MORT_GENERIC_WAVES 4
#endif
#ifndef MORT_MEGA_LDS_LEVELS
#define MORT_MEGA_LDS_LEVELS 9
#endif
extern "C" __global__ void __launch_bounds__(256, MORT_GENERIC_WAVES)
mega_kernel(const RenderArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const int tiles_x = (a.width + 7) >> 3;
    const int tx = wave % tiles_x, ty = wave / tiles_x;
    const int x = tx * 8 + (lane & 7);
    const int ly = ty * 8 + (lane >> 3);
    const bool active = (x < a.width) && (ly < a.local_rows);
    if (!active) return;

    __shared__ float4 s_stack[MORT_MEGA_LDS_LEVELS * 256]; /* the first bounce-stack levels of the block's 256 lanes: 36 KB, four blocks per CU */
    const PixelTotals t = render_pixel<false>(a, nullptr, x, ly, nullptr, s_stack + threadIdx.x, MORT_MEGA_LDS_LEVELS, 256); /* dev_pixel.h: the body the host loop runs too */
    atomicAdd(&a.counters[0], (unsigned long long)t.segments);
    atomicAdd(&a.counters[1], (unsigned long long)t.draws);
}

/* ---- seeding: curand_init(seed, subsequence, 0) ---- */
struct SeedArgs {
    mort_rng_state *states;
    const uint32_t *mats; /* [levels][160][5]: A^(2^67 * 4^k), row i = image of state bit i */
    int levels;
    int width, local_rows, rank, nranks, rows_per_block;
    uint32_t d0, v0, v1, v2, v3, v4; /* scrambled seed */
    int sub; /* 0: the reference's keying, subsequence x + y*W.  S > 0 (MORT_MODE_THROUGHPUT): states[x + (ly*S + j)*W] gets subsequence x + (y*S + j)*W */
};

extern "C" __global__ void __launch_bounds__(256)
seed_kernel(const SeedArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int sub = a.sub > 0 ? a.sub : 1;
    const long long n = (long long)a.width * a.local_rows * sub;
    if (i >= n) return;
    const int vly = i / a.width, x = i - vly * a.width;
    const int ly = vly / sub, j = vly - ly * sub;
    const int y = global_row(ly, a.rank, a.nranks, a.rows_per_block);
    unsigned long long p = (unsigned long long)x + ((unsigned long long)y * (unsigned long long)sub + (unsigned long long)j) * (unsigned long long)a.width;
    uint32_t v[5] = {a.v0, a.v1, a.v2, a.v3, a.v4};
    for (int k = 0; p != 0 && k < a.levels; k++, p >>= 2) {
        const uint32_t *m = a.mats + (size_t)k * 160 * 5;
        const int reps = (int)(p & 3ull);
        for (int t = 0; t < reps; t++) {
            uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0;
            for (int wv = 0; wv < 5; wv++) {
                const uint32_t word = v[wv];
                for (int b = 0; b < 32; b++) {
                    const uint32_t mask = 0u - ((word >> b) & 1u);
                    const uint32_t *row = m + (wv * 32 + b) * 5;
                    r0 ^= row[0] & mask; r1 ^= row[1] & mask; r2 ^= row[2] & mask; r3 ^= row[3] & mask; r4 ^= row[4] & mask;
                }
            }
            v[0] = r0; v[1] = r1; v[2] = r2; v[3] = r3; v[4] = r4;
        }
    }
    mort_rng_state st;
    st.d = a.d0; st.v[0] = v[0]; st.v[1] = v[1]; st.v[2] = v[2]; st.v[3] = v[3]; st.v[4] = v[4];
    st.boxmuller_flag = 0; st.boxmuller_flag_double = 0; st.boxmuller_extra = 0.f; st.boxmuller_extra_double = 0.;
    a.states[i] = st;
}

#include "mega_bvh.h"
#include "wave_bvh.h"

/* finishes the pixels of a sub-stream launch: stratum rows summed in order, then Camera::render's tail (camera.cuh:194-207) */
static __global__ void __launch_bounds__(256) substream_resolve_kernel(const float *vaccum, int width, int local_rows, int sub, float scale, uchar4 *rgba, float *accum) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= width * local_rows) return;
    const int ly = i / width, x = i - ly * width;
    V3 c = mk(0, 0, 0);
    for (int j = 0; j < sub; j++) {
        const size_t v = 3 * ((size_t)(ly * sub + j) * (size_t)width + (size_t)x);
        c = vadd(c, mk(vaccum[v], vaccum[v + 1], vaccum[v + 2]));
    }
    c = vscale(scale, c);
    if (c.x != c.x) c.x = 0.0f;
    if (c.y != c.y) c.y = 0.0f;
    if (c.z != c.z) c.z = 0.0f;
    if (accum) { accum[3 * i] = c.x; accum[3 * i + 1] = c.y; accum[3 * i + 2] = c.z; }
    float g[3] = {mort_sqrtf(c.x), mort_sqrtf(c.y), mort_sqrtf(c.z)};
    unsigned char b[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float v = g[k];
        if (v < 0.0f) v = 0.0f;
        if (v > 0.999f) v = 0.999f;
        b[k] = (unsigned char)mort_f2i(256 * v);
    }
    uchar4 out; out.x = b[0]; out.y = b[1]; out.z = b[2]; out.w = 255;
    rgba[i] = out;
}


/* ====================================================================== host */


/* Wait for everything this context has launched: its own stream and, if a render went to a caller's stream, that one
 * too -- before any call that reads, overwrites or frees what a render kernel uses (states, scene, counters). */
static hipError_t quiesce(mort_ctx *c) {
    hipError_t e = hipSuccess;
    if (c->stream) e = hipStreamSynchronize(c->stream);
    if (c->last_stream && c->last_stream != c->stream) { hipError_t e2 = hipStreamSynchronize(c->last_stream); if (e == hipSuccess) e = e2; }
    c->last_stream = nullptr;
    return e;
}

extern "C" const char *mort_hip_strerror(int st) {
    switch (st) {
    case MORT_OK: return "ok";
    case MORT_ERR_INVALID: return "invalid argument";
    case MORT_ERR_NO_DEVICE: return "no usable HIP device (gfx950 required)";
    case MORT_ERR_HIP: return "HIP runtime error";
    case MORT_ERR_NO_WORLD: return "no world uploaded";
    case MORT_ERR_NO_RNG: return "RNG states not seeded/loaded for this image size";
    case MORT_ERR_UNSUPPORTED: return "scene graph / mode not supported by the kernels";
    case MORT_ERR_CAPACITY: return "capacity exceeded";
    case MORT_ERR_NOMEM: return "out of memory";
    }
    return "unknown status";
}
extern "C" const char *mort_hip_last_error(const mort_ctx *c) { return c ? c->last_error.c_str() : ""; }

static int local_rows_for(const mort_partition &p, int height) {
    int n = 0;
    const int rpb = p.rows_per_block;
    const int nblocks = (height + rpb - 1) / rpb;
    for (int b = p.rank; b < nblocks; b += p.nranks) {
        int r0 = b * rpb, r1 = r0 + rpb;
        if (r1 > height) r1 = height;
        n += r1 - r0;
    }
    return n;
}
static int global_row_host(const mort_partition &p, int ly) {
    const int lb = ly / p.rows_per_block, within = ly % p.rows_per_block;
    return (lb * p.nranks + p.rank) * p.rows_per_block + within;
}

extern "C" int mort_hip_local_rows(const mort_ctx *c, int height) { return c ? local_rows_for(c->part, height) : 0; }
extern "C" int mort_hip_global_row(const mort_ctx *c, int ly) { return c ? global_row_host(c->part, ly) : 0; }

extern "C" int mort_hip_init(int device, mort_ctx **out) {
    if (!out) return MORT_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return MORT_ERR_NO_DEVICE;
    mort_ctx *c = new (std::nothrow) mort_ctx;
    if (!c) return MORT_ERR_NOMEM;
    c->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete c; return MORT_ERR_NO_DEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { delete c; return MORT_ERR_NO_DEVICE; }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) { delete c; return MORT_ERR_NO_DEVICE; }
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipMalloc(&c->d_counters, 96 * sizeof(unsigned long long)) != hipSuccess) {
        mort_hip_shutdown(c);
        return MORT_ERR_HIP;
    }
    *out = c;
    return MORT_OK;
}

extern "C" void mort_hip_shutdown(mort_ctx *c) {
    if (!c) return;
    hipSetDevice(c->device);
    quiesce(c);
    mort_hip_comm_destroy(c);
    hipFree(c->d_tile_keys); hipFree(c->d_tile_iota); hipFree(c->d_sort_tmp);
    hipFree(c->d_prio_count); hipFree(c->d_scene); hipFree(c->d_fast); hipFree(c->d_trav); hipFree(c->d_gen); hipFree(c->d_states); hipFree(c->d_seqmats);
    hipFree(c->d_substates); hipFree(c->d_vaccum);
    hipFree(c->d_rgba); hipFree(c->d_accum); hipFree(c->d_segpx); hipFree(c->d_counters); hipFree(c->d_wf);
    hipFree(c->d_tile_cost); hipFree(c->d_tile_order); hipFree(c->d_probe_states); hipFree(c->d_deep); hipFree(c->d_wave_log);
    if (c->h_live) hipHostFree(c->h_live);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int mort_hip_set_partition(mort_ctx *c, const mort_partition *p) {
    if (!c || !p) return MORT_ERR_INVALID;
    if (p->nranks < 1 || p->rank < 0 || p->rank >= p->nranks || p->rows_per_block < 8 || (p->rows_per_block % 8) != 0)
        return MORT_ERR_INVALID;
    hipSetDevice(c->device);
    quiesce(c);
    c->part = *p;
    c->cost_key = 0;
    /* RNG states are laid out per partition: force a re-seed */
    c->rng_w = c->rng_h = c->rng_local_rows = 0;
    c->sub_w = c->sub_h = c->sub_lr = c->sub_s = 0;
    return MORT_OK;
}

extern "C" int mort_hip_upload_world(mort_ctx *c, const mort_world *w) {
    if (!c || !w) return MORT_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, quiesce(c));
    c->cost_key = 0; c->world_serial++;
    /* nothing of the previous world survives a failed upload: a later render must see MORT_ERR_NO_WORLD, not the old world's
     * trees walked against the new scene tables */
    c->have_world = c->fast_ok = c->gen_ok = c->wave_ok = false;
    if (c->d_fast) { hipFree(c->d_fast); c->d_fast = nullptr; }
    if (c->d_trav) { hipFree(c->d_trav); c->d_trav = nullptr; }
    if (c->d_gen) { hipFree(c->d_gen); c->d_gen = nullptr; }
    int st;
    SceneBlob sb;
    st = build_scene_blob(w, sb);
    if (st != MORT_OK) return st;
    const mortc::Compiled &o = sb.comp;
    void *d = nullptr;
    HIPCHK(c, hipMalloc(&d, sb.bytes.size()));
    hipError_t e = hipMemcpy(d, sb.bytes.data(), sb.bytes.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { hipFree(d); return hip_fail(c, e, "hipMemcpy(scene)"); }
    if (c->d_scene) hipFree(c->d_scene);
    c->d_scene = d;
    scene_view(sb, (const unsigned char *)d, c->sc);
    c->list_types = o.list_types; c->list_idxs = o.list_idxs;
    for (int i = 0; i < MORT_NUM_HITTABLE_LIST; i++) { c->list_first[i] = o.list_first[i]; c->list_count[i] = o.list_count[i]; }
    c->n_wspheres = (int)o.wspheres.size(); c->n_wquads = (int)o.wquads.size(); c->n_lists = w->objs.num_hittable_list;
    /* the LDS kernels handle: one BVH over spheres as the whole world */
    c->wave_ok = (o.items.size() == 1 && o.items[0].kind == ITEM_BVH && o.quads.empty());
    if (c->wave_ok && !o.own_nodes.empty() && !o.own_nodes4.empty()) {
        /* two LDS images: the BVH megakernel's (four-wide nodes, leaf records with their spheres by value, every small
         * table) and the wavefront traversal kernel's (binary nodes, leaf nodes, spheres) */
        std::vector<unsigned char> fb, tb;
        c->f_nodes4 = (uint32_t)place(fb, o.own_nodes4); c->f_leafrecs = (uint32_t)place(fb, o.own_leafrecs);
        c->f_lambert = (uint32_t)place(fb, o.lambert); c->f_metal = (uint32_t)place(fb, o.metal); c->f_diel = (uint32_t)place(fb, o.dielectric);
        c->f_dlight = (uint32_t)place(fb, o.dlight); c->f_iso = (uint32_t)place(fb, o.isotropic);
        c->f_solid = (uint32_t)place(fb, o.solid); c->f_checker = (uint32_t)place(fb, o.checker);
        fb.resize((fb.size() + 15) & ~(size_t)15, 0);
        c->t_nodes2 = (uint32_t)place(tb, o.own_nodes); c->t_leaves = (uint32_t)place(tb, o.own_leaves); c->t_spheres = (uint32_t)place(tb, o.spheres);
        tb.resize((tb.size() + 15) & ~(size_t)15, 0);
        if (fb.size() <= 72 * 1024 && tb.size() <= 72 * 1024) {
            HIPCHK(c, hipMalloc(&c->d_fast, fb.size()));
            HIPCHK(c, hipMemcpy(c->d_fast, fb.data(), fb.size(), hipMemcpyHostToDevice));
            HIPCHK(c, hipMalloc(&c->d_trav, tb.size()));
            HIPCHK(c, hipMemcpy(c->d_trav, tb.data(), tb.size(), hipMemcpyHostToDevice));
            c->fast_bytes = (uint32_t)fb.size(); c->trav_bytes = (uint32_t)tb.size();
            c->own_nodes = (int)o.own_nodes.size(); c->own_leaves = (int)o.own_leaves.size(); c->own4_stack = o.own4_stack > 1 ? o.own4_stack : 1;
            c->fast_ok = true;
        }
    }
    /* unified-tree megakernel: LDS image = tree + every small table (+ the primitives when they fit) */
    if (o.g_ok && !std::getenv("MORT_NO_GEN")) {
        std::vector<unsigned char> gb;
        GenArgs g;
        std::memset(&g, 0, sizeof g);
        g.o_nodes = (uint32_t)place(gb, o.g_nodes); g.o_leaves = g.o_nodes; g.o_entries = (uint32_t)place(gb, o.g_entries);
        g.o_chains = (uint32_t)place(gb, o.g_chains); g.o_xforms = (uint32_t)place(gb, o.xforms);
        g.o_items = (uint32_t)place(gb, o.items); g.o_subitems = (uint32_t)place(gb, o.subitems); g.o_media = (uint32_t)place(gb, o.media);
        g.o_lambert = (uint32_t)place(gb, o.lambert); g.o_metal = (uint32_t)place(gb, o.metal); g.o_diel = (uint32_t)place(gb, o.dielectric);
        g.o_dlight = (uint32_t)place(gb, o.dlight); g.o_iso = (uint32_t)place(gb, o.isotropic);
        g.o_solid = (uint32_t)place(gb, o.solid); g.o_checker = (uint32_t)place(gb, o.checker); g.o_image = (uint32_t)place(gb, o.image);
        g.o_lfirst = (uint32_t)place(gb, std::vector<int>(o.list_first, o.list_first + MORT_NUM_HITTABLE_LIST));
        g.o_lcount = (uint32_t)place(gb, std::vector<int>(o.list_count, o.list_count + MORT_NUM_HITTABLE_LIST));
        const size_t prim_bytes = o.spheres.size() * sizeof(DSphere) + o.quads.size() * sizeof(DQuad) + o.wspheres.size() * sizeof(DSphere) +
                                  o.wquads.size() * sizeof(DQuad) + (o.list_types.size() + o.list_idxs.size()) * sizeof(int) + 6 * 32;
        if (gb.size() + prim_bytes <= 48 * 1024) {
            g.prims_in_lds = 1;
            g.o_spheres = (uint32_t)place(gb, o.spheres); g.o_quads = (uint32_t)place(gb, o.quads);
            g.o_wspheres = (uint32_t)place(gb, o.wspheres); g.o_wquads = (uint32_t)place(gb, o.wquads);
            g.o_ltypes = (uint32_t)place(gb, o.list_types); g.o_lidxs = (uint32_t)place(gb, o.list_idxs);
        }
        gb.resize((gb.size() + 15) & ~(size_t)15, 0);
        const size_t lds_part = gb.size();
        const size_t o_ranks = place(gb, o.g_ranks); /* HBM only: read when two hits have equal t */
#ifndef MORT_GEN_IMAGE_MAX
#define MORT_GEN_IMAGE_MAX (124 * 1024) /* + traversal stacks of 768 threads (24 KB) + the launch arguments: one workgroup per CU */
#endif
        if (lds_part <= MORT_GEN_IMAGE_MAX) {
            HIPCHK(c, hipMalloc(&c->d_gen, gb.size()));
            HIPCHK(c, hipMemcpy(c->d_gen, gb.data(), gb.size(), hipMemcpyHostToDevice));
            c->gen_bytes = (uint32_t)lds_part;
            g.ranks = (const uint32_t *)((const unsigned char *)c->d_gen + o_ranks); g.n_spheres = (int)o.spheres.size();
            g.root = o.g_root; g.first_medium = o.g_first_medium; g.n_chains = (int)(o.g_chains.size() / 2);
            g.gx = o.g_c[0]; g.gy = o.g_c[1]; g.gz = o.g_c[2]; g.gR = o.g_R; g.mnear = o.g_mnear; g.kmin = o.g_kmin;
            c->gen = g;
            for (int k = 0; k < 3; k++) { c->gen_lo[k] = o.g_lo[k]; c->gen_hi[k] = o.g_hi[k]; }
            c->gen_reach = o.g_reach;
            c->gen_prims = (int)o.g_entries.size();
            c->gen_ok = true;
        }
    }
    c->have_world = true;
    return MORT_OK;
}

static int ensure_states(mort_ctx *c, int width, int height) {
    const int lr = local_rows_for(c->part, height);
    if (c->d_states && c->rng_w == width && c->rng_h == height && c->rng_local_rows == lr) return MORT_OK;
    if (c->d_states) { hipFree(c->d_states); c->d_states = nullptr; }
    c->rng_w = c->rng_h = c->rng_local_rows = 0;
    size_t n = (size_t)width * (size_t)(lr > 0 ? lr : 1);
    HIPCHK(c, hipMalloc((void **)&c->d_states, n * sizeof(mort_rng_state)));
    c->rng_w = width; c->rng_h = height; c->rng_local_rows = lr;
    return MORT_OK;
}

extern "C" int mort_hip_rng_seed(mort_ctx *c, uint64_t seed, int width, int height) {
    if (!c || width <= 0 || height <= 0) return MORT_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->d_seqmats) {
        std::vector<XMat> seq;
        build_seq_matrices(seq);
        HIPCHK(c, hipMalloc((void **)&c->d_seqmats, seq.size() * sizeof(XMat)));
        HIPCHK(c, hipMemcpy(c->d_seqmats, seq.data(), seq.size() * sizeof(XMat), hipMemcpyHostToDevice));
    }
    HIPCHK(c, quiesce(c));
    int st = ensure_states(c, width, height);
    if (st != MORT_OK) return st;
    SeedArgs a;
    a.states = c->d_states; a.mats = c->d_seqmats; a.levels = SEQ_LEVELS;
    a.width = width; a.local_rows = c->rng_local_rows;
    a.rank = c->part.rank; a.nranks = c->part.nranks; a.rows_per_block = c->part.rows_per_block;
    const SeedWords sw = seed_scramble(seed);
    a.d0 = sw.d; a.v0 = sw.v[0]; a.v1 = sw.v[1]; a.v2 = sw.v[2]; a.v3 = sw.v[3]; a.v4 = sw.v[4];
    a.sub = 0;
    c->seed = seed; c->seed_known = true;
    c->sub_w = c->sub_h = c->sub_lr = c->sub_s = 0; /* a re-seed restarts the sub-streams too */
    const int n = width * c->rng_local_rows;
    if (n > 0) {
        hipLaunchKernelGGL(seed_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, a);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return MORT_OK;
}

extern "C" int mort_hip_rng_load(mort_ctx *c, const mort_rng_state *states, int width, int height) {
    if (!c || !states || width <= 0 || height <= 0) return MORT_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, quiesce(c));
    int st = ensure_states(c, width, height);
    if (st != MORT_OK) return st;
    c->seed_known = false;
    c->sub_w = c->sub_h = c->sub_lr = c->sub_s = 0;
    for (int ly = 0; ly < c->rng_local_rows; ly++) {
        const int y = global_row_host(c->part, ly);
        HIPCHK(c, hipMemcpy(c->d_states + (size_t)ly * width, states + (size_t)y * width, (size_t)width * sizeof(mort_rng_state), hipMemcpyHostToDevice));
    }
    return MORT_OK;
}

extern "C" int mort_hip_rng_store(mort_ctx *c, mort_rng_state *states, int width, int height) {
    if (!c || !states || width <= 0 || height <= 0) return MORT_ERR_INVALID;
    if (!c->d_states || c->rng_w != width || c->rng_h != height) return MORT_ERR_NO_RNG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, quiesce(c));
    for (int ly = 0; ly < c->rng_local_rows; ly++) {
        const int y = global_row_host(c->part, ly);
        HIPCHK(c, hipMemcpy(states + (size_t)y * width, c->d_states + (size_t)ly * width, (size_t)width * sizeof(mort_rng_state), hipMemcpyDeviceToHost));
    }
    return MORT_OK;
}


static int check_light(const mort_ctx *c, int type, int idx) {
    if (type == -1) return MORT_OK;
    if (type == MORT_OBJ_SPHERE) return (idx >= 0 && idx < c->n_wspheres) ? MORT_OK : MORT_ERR_INVALID;
    if (type == MORT_OBJ_QUAD) return (idx >= 0 && idx < c->n_wquads) ? MORT_OK : MORT_ERR_INVALID;
    if (type == MORT_OBJ_HITTABLE_LIST) {
        if (idx < 0 || idx >= c->n_lists || idx >= MORT_NUM_HITTABLE_LIST) return MORT_ERR_INVALID;
        if (c->list_count[idx] <= 0) return MORT_ERR_INVALID;
        for (int i = 0; i < c->list_count[idx]; i++) {
            const int t = c->list_types[c->list_first[idx] + i], k = c->list_idxs[c->list_first[idx] + i];
            if (t == MORT_OBJ_SPHERE) { if (k < 0 || k >= c->n_wspheres) return MORT_ERR_INVALID; }
            else if (t == MORT_OBJ_QUAD) { if (k < 0 || k >= c->n_wquads) return MORT_ERR_INVALID; }
            else if (t == MORT_OBJ_HITTABLE_LIST) return MORT_ERR_UNSUPPORTED; /* nested light lists */
        }
        return MORT_OK;
    }
    return MORT_OK; /* any other tag samples nothing: pdf 0, direction (1,0,0) (objects.cuh:961,978) */
}

/* ---- wavefront mode: one wf_trav + one wf_shade launch per front until no pixel is live ---- */
static int render_wavefront(mort_ctx *c, const RenderArgs &a, const mort_camera *cam, hipStream_t s) {
    const size_t N = (size_t)a.width * (size_t)a.local_rows;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
    const size_t o_ray0 = take(N * sizeof(WfRay)), o_ray1 = take(N * sizeof(WfRay));
    const size_t o_id0 = take(N * sizeof(unsigned)), o_id1 = take(N * sizeof(unsigned));
    const size_t o_hits = take(N * sizeof(WfHit)), o_pix = take(N * sizeof(WfPix));
    const size_t o_stack = take(N * (size_t)cam->bounce_limit * sizeof(float4));
    const size_t o_c0 = take(N * sizeof(unsigned)), o_c1 = take(N * sizeof(unsigned)), o_c2 = take(N * sizeof(unsigned));
    const size_t o_cnt = take(sizeof(WfCounters));
    const size_t total = off;
    if (c->wf_bytes < total) {
        if (c->d_wf) { hipFree(c->d_wf); c->d_wf = nullptr; c->wf_bytes = 0; }
        HIPCHK(c, hipMalloc(&c->d_wf, total));
        c->wf_bytes = total;
    }
    if (!c->h_live) HIPCHK(c, hipHostMalloc((void **)&c->h_live, 64));
    unsigned char *base = (unsigned char *)c->d_wf;
    WfArgs w;
    std::memset(&w, 0, sizeof w);
    w.r = a;
    w.node_first = 0; w.node_count = c->sc.n_nodes;
    w.n_paths = (int)N;
    w.q_ray[0] = (WfRay *)(base + o_ray0); w.q_ray[1] = (WfRay *)(base + o_ray1);
    w.q_id[0] = (unsigned *)(base + o_id0); w.q_id[1] = (unsigned *)(base + o_id1);
    w.hits = (WfHit *)(base + o_hits); w.pix = (WfPix *)(base + o_pix);
    w.stack = (float4 *)(base + o_stack);
    w.q_cls[0] = (unsigned *)(base + o_c0); w.q_cls[1] = (unsigned *)(base + o_c1); w.q_cls[2] = (unsigned *)(base + o_c2);
    w.cnt = (WfCounters *)(base + o_cnt);

    const int nb256 = (int)((N + 255) / 256);
    hipLaunchKernelGGL(wf_init, dim3(nb256), dim3(256), 0, s, w);
    HIPCHK(c, hipGetLastError());
    auto trav = wf_trav<MORT_WF_BLOCK>;
    const size_t stage_bytes = (size_t)(MORT_WF_BLOCK / 64) * 3 * MORT_WF_STAGE * sizeof(unsigned);
    /* wf_trav's LDS: its image (binary own tree, leaf nodes, spheres) | traversal stacks | class staging | prefetch rings */
    w.trav_src = (const unsigned char *)c->d_trav; w.trav_bytes = c->trav_bytes;
    w.t_nodes2 = c->t_nodes2; w.t_leaves = c->t_leaves; w.t_spheres = c->t_spheres;
    w.t_tstack = (c->trav_bytes + 15u) & ~15u;
    w.t_stage = w.t_tstack + (uint32_t)MORT_OWN_STACK * (uint32_t)MORT_WF_BLOCK * 2u;
    const size_t ring_off = (((size_t)w.t_stage + stage_bytes) + 1023) & ~(size_t)1023;
    const size_t trav_lds = ring_off + (size_t)(MORT_WF_BLOCK / 64) * 4096;
    w.off_ring = (uint32_t)ring_off;
    HIPCHK(c, hipFuncSetAttribute((const void *)trav, hipFuncAttributeMaxDynamicSharedMemorySize, (int)trav_lds));
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, trav, MORT_WF_BLOCK, trav_lds) != hipSuccess || per_cu < 1) per_cu = 1;
    const int max_trav_grid = c->num_cus * per_cu;
    size_t wf_share = 1; /* 64-record batches per wave at least (MORT_WAVE_SHARE; 2 was 4 % slower: a front's time is its slowest wave's) */
    { const char *sh = std::getenv("MORT_WAVE_SHARE"); if (sh && std::atoi(sh) >= 1) wf_share = (size_t)std::atoi(sh); }
    const long long max_fronts = (long long)cam->sqrt_spp * cam->sqrt_spp * ((long long)cam->bounce_limit + 1) + 8;
    size_t live = N;
    long long front = 0;
    const int chunk = 32;
    while (live > 0 && front < max_fronts) {
        int tg = (int)((live + (size_t)(MORT_WF_BLOCK / 64) * 64 * wf_share - 1) / ((size_t)(MORT_WF_BLOCK / 64) * 64 * wf_share));
        if (tg > max_trav_grid) tg = max_trav_grid;
        if (tg < 1) tg = 1;
        const int sg = (int)((live + 255) / 256) + 3;
        for (int k = 0; k < chunk; k++, front++) {
            w.parity = (int)(front & 1);
            hipLaunchKernelGGL(trav, dim3(tg), dim3(MORT_WF_BLOCK), trav_lds, s, w);
            hipLaunchKernelGGL(wf_shade, dim3(sg), dim3(256), 0, s, w);
        }
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(c->h_live, &w.cnt->live, sizeof(unsigned), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        live = *c->h_live;
    }
    c->wf_fronts = (int)front;
    if (live != 0) { c->last_error = "wavefront: front limit reached with live pixels"; return MORT_ERR_HIP; }
    return MORT_OK;
}

/* ---- tile order of the state-machine megakernels: most expensive 8x8 tiles first (cost = segments per tile in the
 * previous frame of this world / view / partition, or in a one-sample probe on a scratch copy of the streams), so a frame
 * does not end on its longest pixel chains.  Everything is queued on `s`; nothing waits on the host. ---- */
template <typename ProbeFn>
static int prepare_tile_order(mort_ctx *c, const mort_camera *cam, const RenderArgs &a, FastArgs &fa, int tiles, int grid, int FB,
                              bool chain_bound, hipStream_t s, ProbeFn launch_probe) {
    const int W = a.width, H = a.height;
    if (c->tile_cap < (size_t)tiles) {
        hipFree(c->d_tile_cost); hipFree(c->d_tile_order); hipFree(c->d_tile_keys); hipFree(c->d_tile_iota); hipFree(c->d_sort_tmp);
        c->d_tile_cost = c->d_tile_order = c->d_tile_keys = c->d_tile_iota = nullptr; c->d_sort_tmp = nullptr;
        c->tile_cap = 0; c->sort_tmp_bytes = 0;
        HIPCHK(c, hipMalloc((void **)&c->d_tile_cost, (size_t)tiles * sizeof(unsigned)));
        HIPCHK(c, hipMalloc((void **)&c->d_tile_order, (size_t)tiles * sizeof(unsigned)));
        HIPCHK(c, hipMalloc((void **)&c->d_tile_keys, (size_t)tiles * sizeof(unsigned)));
        HIPCHK(c, hipMalloc((void **)&c->d_tile_iota, (size_t)tiles * sizeof(unsigned)));
        c->sort_tmp_bytes = mort_tile_sort_temp_bytes(tiles);
        HIPCHK(c, hipMalloc(&c->d_sort_tmp, c->sort_tmp_bytes ? c->sort_tmp_bytes : 16));
        c->tile_cap = (size_t)tiles;
        c->cost_key = 0;
    }
    if (fa.heavy_mod > 0 && !c->d_prio_count) {
        HIPCHK(c, hipMalloc((void **)&c->d_prio_count, 16 + 96 * sizeof(unsigned long long)));
        HIPCHK(c, hipMemsetAsync(c->d_prio_count, 0, 16 + 96 * sizeof(unsigned long long), s));
    }
    /* the costs belong to one (world, image geometry, partition, view): FNV-1a over all of it */
    unsigned long long key = 1469598103934665603ull;
    auto mix = [&key](const void *p, size_t n) { const unsigned char *b = (const unsigned char *)p; for (size_t i = 0; i < n; i++) { key ^= b[i]; key *= 1099511628211ull; } };
    { const int g[9] = {W, H, a.local_rows, a.rank, a.nranks, a.rows_per_block, cam->bounce_limit, (int)c->world_serial, a.light_type * 65536 + a.light_idx}; mix(g, sizeof g); }
    mix(&a.center, sizeof a.center); mix(&a.pixel00, sizeof a.pixel00); mix(&a.du, sizeof a.du); mix(&a.dv, sizeof a.dv);
    mix(&a.defocus_angle, sizeof a.defocus_angle);
    if (key == 0) key = 1;
    if (c->cost_key != key) { /* no history for this view: one-sample probe on a scratch copy of the streams */
        const size_t npx = (size_t)W * (size_t)a.local_rows;
        if (c->probe_cap < npx) {
            if (c->d_probe_states) { hipFree(c->d_probe_states); c->d_probe_states = nullptr; c->probe_cap = 0; }
            HIPCHK(c, hipMalloc((void **)&c->d_probe_states, npx * sizeof(mort_rng_state)));
            c->probe_cap = npx;
        }
        HIPCHK(c, hipMemcpyAsync(c->d_probe_states, c->d_states, npx * sizeof(mort_rng_state), hipMemcpyDeviceToDevice, s));
        HIPCHK(c, hipMemsetAsync(c->d_tile_cost, 0, (size_t)tiles * sizeof(unsigned), s));
        FastArgs pa = fa;
        pa.r.states = c->d_probe_states; pa.r.sqrt_spp = 1; pa.r.recip_sqrt_spp = 1.0f; pa.r.pixel_samples_scale = 1.0f;
        pa.r.accum = nullptr; pa.r.seg_px = nullptr;
        pa.tile_order = nullptr; pa.tile_cost = c->d_tile_cost;
        { const char *tk = std::getenv("MORT_TILE_KEY"); pa.tile_key_sum = (tk && std::strcmp(tk, "sum") == 0) ? 1 : 0; }
        HIPCHK(c, launch_probe(pa));
        if (c->d_prio_count) HIPCHK(c, hipMemcpyAsync(c->d_prio_count + 4, c->d_counters, 96 * sizeof(unsigned long long), hipMemcpyDeviceToDevice, s)); /* the probe's counters (its segment total), beside its tile costs */
        HIPCHK(c, hipMemsetAsync(c->d_counters, 0, 96 * sizeof(unsigned long long), s)); /* probe totals and work cursor */
        c->cost_key = key;
    }
    /* order = argsort(cost, descending, equal costs by index), on the device and on this stream */
    HIPCHK(c, mort_tile_sort_desc(c->d_tile_cost, c->d_tile_keys, c->d_tile_iota, c->d_tile_order, c->d_sort_tmp, c->sort_tmp_bytes, tiles, s));
    /* the head of the order: tiles whose longest pixel reaches heavy_percent % of the frame's longest (unified-tree kernel's heavy waves) */
    fa.prio_dev = nullptr;
    if (fa.heavy_mod > 0) {
        /* d_prio_count: [0] the head's size for this launch (u32); from byte 16: the 96 counters of the frame (or probe) the tile costs come from */
        HIPCHK(c, mort_tile_heavy_count(c->d_tile_keys, tiles, (unsigned)c->heavy_percent, (unsigned)(tiles / 2 > 0 ? tiles / 2 : 1),
                                        (const unsigned long long *)(c->d_prio_count + 4), (unsigned long long)grid * (unsigned long long)FB, c->d_prio_count, s));
        fa.prio_dev = c->d_prio_count;
    }
    HIPCHK(c, hipMemsetAsync(c->d_tile_cost, 0, (size_t)tiles * sizeof(unsigned), s));
    fa.tile_order = c->d_tile_order; fa.tile_cost = c->d_tile_cost;
    { const char *tk = std::getenv("MORT_TILE_KEY"); fa.tile_key_sum = (tk && std::strcmp(tk, "sum") == 0) ? 1 : 0; }
    fa.gen_tiles = grid * (FB / 64); /* one tile's worth of slots per wave in flight */
    fa.spread_shift = chain_bound ? 0 : 6; /* measured: whole tiles while lanes refill several times, single pixels otherwise */
    { const char *sp = std::getenv("MORT_SPREAD_SHIFT"); if (sp) fa.spread_shift = std::atoi(sp); }
    if (fa.spread_shift >= 6 || fa.spread_shift < 0) fa.gen_tiles = 0;
    return MORT_OK;
}

/* MORT_MODE_THROUGHPUT: the (pixel, stratum row) streams for this image, partition and sqrt_spp; seeded once, then carried from
 * frame to frame like the per-pixel states */
static int ensure_substates(mort_ctx *c, int W, int H, int S, hipStream_t s) {
    const int lr = c->rng_local_rows;
    const size_t n = (size_t)W * (size_t)lr * (size_t)S;
    if (n >= (1ull << 31)) return MORT_ERR_CAPACITY;
    if (c->substates_cap < n) {
        if (c->d_substates) { hipFree(c->d_substates); c->d_substates = nullptr; }
        c->substates_cap = 0; c->sub_s = 0;
        HIPCHK(c, hipMalloc((void **)&c->d_substates, (n ? n : 1) * sizeof(mort_rng_state)));
        c->substates_cap = n;
    }
    if (c->vaccum_cap < n) {
        if (c->d_vaccum) { hipFree(c->d_vaccum); c->d_vaccum = nullptr; }
        c->vaccum_cap = 0;
        HIPCHK(c, hipMalloc((void **)&c->d_vaccum, (n ? n : 1) * 3 * sizeof(float)));
        c->vaccum_cap = n;
    }
    if (c->sub_w == W && c->sub_h == H && c->sub_lr == lr && c->sub_s == S) return MORT_OK;
    SeedArgs a;
    a.states = c->d_substates; a.mats = c->d_seqmats; a.levels = SEQ_LEVELS;
    a.width = W; a.local_rows = lr;
    a.rank = c->part.rank; a.nranks = c->part.nranks; a.rows_per_block = c->part.rows_per_block;
    const SeedWords sw = seed_scramble(c->seed);
    a.d0 = sw.d; a.v0 = sw.v[0]; a.v1 = sw.v[1]; a.v2 = sw.v[2]; a.v3 = sw.v[3]; a.v4 = sw.v[4];
    a.sub = S;
    if (n > 0) {
        hipLaunchKernelGGL(seed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
        HIPCHK(c, hipGetLastError());
    }
    c->sub_w = W; c->sub_h = H; c->sub_lr = lr; c->sub_s = S;
    return MORT_OK;
}

/* d_segpx: per-pixel segment counts for the packed owned rows, or null.  Only mort_hip_render passes one (sized for
 * THIS image and partition); the public device entry never does, so a buffer left over from an earlier, smaller
 * render can not be written past its end. */
static int ensure_buf(mort_ctx *c, void **p, size_t *cap, size_t need);
static int render_device_impl(mort_ctx *c, const mort_camera *cam, int mode, void *d_rgba, void *d_accum, uint32_t *d_segpx,
                              void *stream, mort_stats *stats) {
    if (!c || !cam || !d_rgba) return MORT_ERR_INVALID;
    if (mode != MORT_MODE_MEGA && mode != MORT_MODE_WAVE && mode != MORT_MODE_THROUGHPUT) return MORT_ERR_INVALID;
    if (!c->have_world) return MORT_ERR_NO_WORLD;
    const int W = cam->image_width, H = cam->image_height;
    if (W <= 0 || H <= 0 || cam->sqrt_spp < 0) return MORT_ERR_INVALID;
    if (cam->bounce_limit < 0 || cam->bounce_limit > MORT_MAX_BOUNCE_LIMIT) return MORT_ERR_CAPACITY;
    if (!c->d_states || c->rng_w != W || c->rng_h != H) return MORT_ERR_NO_RNG;
    int st = check_light(c, cam->light_obj_type, cam->light_obj_idx);
    if (st != MORT_OK) return st;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    /* a render still running on another stream uses the same states and counters */
    if (c->last_stream && c->last_stream != s) HIPCHK(c, hipStreamSynchronize(c->last_stream));
    c->last_stream = s;

    RenderArgs a;
    std::memset(&a, 0, sizeof a);
    a.sc = c->sc;
    render_args_camera(a, cam);
    a.rank = c->part.rank; a.nranks = c->part.nranks; a.rows_per_block = c->part.rows_per_block;
    a.local_rows = c->rng_local_rows;
    a.states = c->d_states;
    a.rgba = (uchar4 *)d_rgba; a.accum = (float *)d_accum; a.seg_px = d_segpx;
    a.counters = c->d_counters;
    a.debug_lofs = -1;
    { const char *dp = std::getenv("MORT_DEBUG_PIXEL"); if (dp) a.debug_lofs = std::atoi(dp); }

    HIPCHK(c, hipMemsetAsync(c->d_counters, 0, 96 * sizeof(unsigned long long), s));
    /* MORT_MODE_THROUGHPUT: the launch covers a virtual image of local_rows * sqrt_spp rows (mega_bvh.h FastArgs.sub) */
    const bool substream = mode == MORT_MODE_THROUGHPUT;
    const int tiles = ((W + 7) / 8) * (((substream ? a.local_rows * (cam->sqrt_spp > 0 ? cam->sqrt_spp : 1) : a.local_rows) + 7) / 8);
    const int waves_per_block = 4;
    const int blocks = (tiles + waves_per_block - 1) / waves_per_block;
    const char *force = std::getenv("MORT_FORCE_GENERIC");
    const bool use_fast = c->fast_ok && cam->light_obj_type == -1 && cam->sqrt_spp >= 1 && cam->sqrt_spp < 32768 && cam->bounce_limit >= 1 &&
                          W < 65536 && H < 32768 &&
                          !(force && force[0] == '1');
    /* the unified-tree megakernel (mega_gen.hip): worlds without reference BVHs; the camera must lie where the tree's
     * pads were sized for (scene_compile.h build_unified) */
    bool use_gen = c->gen_ok && !use_fast && cam->sqrt_spp >= 1 && cam->sqrt_spp < 32768 && cam->bounce_limit >= 1 && W < 65536 && H < 32768 &&
                   !(force && force[0] == '1');
    /* small worlds stay on the one-lane-per-pixel kernel: scanning a dozen primitives in lockstep keeps every lane busy,
     * a tree walk scheduled by state does not (Cornell box 800x800x100: 53 ms vs 83 ms) */
    if (use_gen) {
        int min_prims = 48;
        if (const char *mp = std::getenv("MORT_GEN_MIN_PRIMS")) min_prims = std::atoi(mp);
        if (c->gen_prims < min_prims) use_gen = false;
    }
    if (use_gen) {
        const float rad = std::fabs(cam->defocus_disk_u.e[0]) + std::fabs(cam->defocus_disk_u.e[1]) + std::fabs(cam->defocus_disk_u.e[2]) +
                          std::fabs(cam->defocus_disk_v.e[0]) + std::fabs(cam->defocus_disk_v.e[1]) + std::fabs(cam->defocus_disk_v.e[2]);
        for (int k = 0; k < 3; k++) {
            const float v = cam->center.e[k];
            if (!(v - rad >= c->gen_lo[k] - c->gen_reach && v + rad <= c->gen_hi[k] + c->gen_reach)) use_gen = false;
        }
    }
    int lds_bytes_used = 0, gen_block_used = 0;
    const void *fast_kernel_used = nullptr;
    char kname[64] = "mega_kernel";
    /* the wavefront pipeline: wave_bvh.h for one reference BVH of spheres as the world (no light object), wave_gen.hip for
     * every world with a unified tree (lights, quads, instances, media); anything else is not supported in this mode */
    const bool wave_bvh = c->wave_ok && c->fast_ok && cam->light_obj_type == -1;
    bool wave_gen = false;
    if (mode == MORT_MODE_WAVE) {
        if (!(cam->sqrt_spp >= 1 && cam->sqrt_spp < 4096 && cam->bounce_limit >= 1)) return MORT_ERR_UNSUPPORTED;
        if (!wave_bvh) {
            wave_gen = c->gen_ok;
            const float rad = std::fabs(cam->defocus_disk_u.e[0]) + std::fabs(cam->defocus_disk_u.e[1]) + std::fabs(cam->defocus_disk_u.e[2]) +
                              std::fabs(cam->defocus_disk_v.e[0]) + std::fabs(cam->defocus_disk_v.e[1]) + std::fabs(cam->defocus_disk_v.e[2]);
            for (int k = 0; k < 3 && wave_gen; k++) {
                const float v = cam->center.e[k];
                if (!(v - rad >= c->gen_lo[k] - c->gen_reach && v + rad <= c->gen_hi[k] + c->gen_reach)) wave_gen = false;
            }
            if (!wave_gen) return MORT_ERR_UNSUPPORTED;
        }
    }
    if (substream) {
        if (!(use_fast || use_gen) || d_segpx) return MORT_ERR_UNSUPPORTED; /* the two LDS state-machine kernels only */
        if (!c->seed_known) return MORT_ERR_NO_RNG;
        if ((long long)a.local_rows * cam->sqrt_spp >= 32768ll * 64) return MORT_ERR_CAPACITY;
        int st_s = ensure_substates(c, W, H, cam->sqrt_spp, s);
        if (st_s != MORT_OK) return st_s;
    }
    if (stats) HIPCHK(c, hipEventRecord(c->ev0, s));
    if (blocks > 0 && mode == MORT_MODE_WAVE && wave_gen) {
        GenArgs ga = c->gen;
        ga.f.r = a;
        ga.f.hot_src = (const unsigned char *)c->d_gen; ga.f.hot_bytes = c->gen_bytes;
        ga.f.th_s = 32; ga.f.th_l = 24; ga.f.t_keep = 16; /* wf_trav_gen's retire+refill / leaf / box-run thresholds */
        { const char *th = std::getenv("MORT_WAVE_THRESHOLDS"); /* "f,l,k" */
          if (th) { int f_ = 0, l_ = 0, k_ = 0; if (std::sscanf(th, "%d,%d,%d", &f_, &l_, &k_) == 3) { ga.f.th_s = f_; ga.f.th_l = l_; ga.f.t_keep = k_; } } }
        if (!c->h_live) HIPCHK(c, hipHostMalloc((void **)&c->h_live, 64));
        WfGenHost hb;
        hb.d_wf = &c->d_wf; hb.wf_bytes = &c->wf_bytes; hb.h_live = &c->h_live; hb.fronts = &c->wf_fronts; hb.num_cus = c->num_cus;
        unsigned live_left = 0;
        hipError_t e_w = mort_wave_gen_render(ga, hb, cam->bounce_limit, cam->sqrt_spp, s, &live_left);
        if (e_w != hipSuccess) {
            if (live_left) { c->last_error = "wavefront: front limit reached with live pixels"; return MORT_ERR_HIP; }
            return hip_fail(c, e_w, "mort_wave_gen_render");
        }
        lds_bytes_used = (int)c->gen_bytes;
        std::snprintf(kname, sizeof kname, "wf_trav_gen<%s>", ga.prims_in_lds ? "true" : "false");
    } else if (blocks > 0 && mode == MORT_MODE_WAVE) {
        int st_w = render_wavefront(c, a, cam, s);
        if (st_w != MORT_OK) return st_w;
        lds_bytes_used = (int)c->trav_bytes;
        std::snprintf(kname, sizeof kname, "wf_trav<%d>", MORT_WF_BLOCK);
    } else if (blocks > 0 && use_fast) {
        FastArgs fa;
        std::memset(&fa, 0, sizeof fa);
        fa.r = a;
        fa.hot_src = (const unsigned char *)c->d_fast; fa.hot_bytes = c->fast_bytes;
        fa.off_nodes4 = c->f_nodes4; fa.off_leafrecs = c->f_leafrecs; fa.off_lambert = c->f_lambert; fa.off_metal = c->f_metal;
        fa.off_diel = c->f_diel; fa.off_dlight = c->f_dlight; fa.off_iso = c->f_iso; fa.off_solid = c->f_solid; fa.off_checker = c->f_checker;
        fa.node_first = 0; fa.node_count = c->sc.n_nodes; /* the reference's threaded nodes (HBM): fallback walk */
        fa.next_q = (unsigned int *)(c->d_counters + 2);
        fa.tiles_x = (W + 7) / 8; fa.tiles_total = tiles;
        if (substream) { fa.sub = cam->sqrt_spp; fa.vaccum = c->d_vaccum; fa.r.states = c->d_substates; }
        const long long lanes_wanted = (long long)tiles * 64;
        const double px_per_lane = (double)lanes_wanted / ((double)c->num_cus * 768.0);
        /* workgroup size.  Two pixels per lane or more: 1024 threads, compiled for 128 registers = 16 waves per CU.  A wave issues a
         * dependent vector instruction every ~8 cycles and an independent one every ~5 (calibration), a SIMD can issue one
         * every ~1.2: a fourth wave per SIMD is worth more than the 24 registers the 1024-thread build keeps in private memory, all of
         * them touched in the shade step only (Scene 1: 103.7 ms against 111.9 ms with 768 threads at 153 registers).
         * Otherwise the largest of {768, 512, 384, 256} that still gives every CU a workgroup
         * (a rank of an 8-way partition owns ~100 k pixels: 768-thread groups would leave half the CUs idle) */
        int FB = MORT_FAST_BLOCK;
        const char *fb_env = std::getenv("MORT_FAST_BLOCK_SIZE");
        if (fb_env) FB = std::atoi(fb_env);
        else {
            const int cand[4] = {768, 512, 384, 256};
            FB = 256;
            const bool wide_ok = lanes_wanted >= 2ll * 1024 * c->num_cus;
            /* about one pixel per lane or fewer: the frame is as long as its longest pixel chain, so take the drain kernels that are
             * compiled without spills (<= 512 threads; one rank of 4 at 1200x675: 81 ms vs 90 ms with 768) */
            for (int k = (px_per_lane < 1.5 && !substream) ? 1 : 0; k < 4; k++) if (lanes_wanted >= (long long)cand[k] * c->num_cus) { FB = cand[k]; break; }
            if (wide_ok) FB = 1024;
        }
        /* 1024 threads: image + traversal stacks + one bounce-stack level per lane must fit one CU's LDS, else the widest shape that does */
        if (FB == 1024 && (size_t)((fa.hot_bytes + 15u) & ~15u) + (size_t)c->own4_stack * 1024u * 2u + 2048u + 1024u * 16u > 160u * 1024u) FB = 768;
        fa.lane_cap = 64;
        { const char *lc = std::getenv("MORT_LANE_CAP"); if (lc && std::atoi(lc) >= 1 && std::atoi(lc) <= 64) fa.lane_cap = std::atoi(lc); }
        fa.drain_rounds = 3; /* batch thresholds as shares of the wave's LIVE lanes (they differ from fixed counts only once lanes have run out of pixels: the tail of a frame;
                              * three runs each, one box: N = 1 100.1-100.5 vs 100.3-102.9 ms, a rank of 2 75.6-78.6 vs 77.6-82.2 ms, ranks of 4 / 8 unchanged); DRAIN kernels
                              * also follow the lane furthest behind (round 2).  MORT_BVH_DRAIN: 0 / 1 = fixed counts, 2 = rounds, 3 = this */
        { const char *dm = std::getenv("MORT_BVH_DRAIN"); if (dm) fa.drain_rounds = std::atoi(dm) == 2 ? 1 : std::atoi(dm) == 3 ? 3 : 0; }
        void (*kern)(const FastArgs) = nullptr, (*kern_probe)(const FastArgs) = nullptr;
        /* chain-bound partition (about one pixel per lane or fewer): drain mode + spread fetches (mega_bvh.h) */
        bool chain_bound = px_per_lane < 1.5 && !substream;
        { const char *cb = std::getenv("MORT_CHAIN_BOUND"); if (cb && !substream) chain_bound = cb[0] == '1'; }
        switch (FB) {
        case 1024: kern = mega_bvh_kernel<1024, false, false>; kern_probe = mega_bvh_kernel<1024, true, false>; chain_bound = false; break; /* no drain variant: chain-bound partitions take <= 512 threads */
        case 768: kern = chain_bound ? mega_bvh_kernel<768, false, true> : mega_bvh_kernel<768, false, false>; kern_probe = mega_bvh_kernel<768, true, false>; break;
        case 512: kern = chain_bound ? mega_bvh_kernel<512, false, true> : mega_bvh_kernel<512, false, false>; kern_probe = mega_bvh_kernel<512, true, false>; break;
        case 384: kern = chain_bound ? mega_bvh_kernel<384, false, true> : mega_bvh_kernel<384, false, false>; kern_probe = mega_bvh_kernel<384, true, false>; break;
        default: FB = 256; kern = chain_bound ? mega_bvh_kernel<256, false, true> : mega_bvh_kernel<256, false, false>; kern_probe = mega_bvh_kernel<256, true, false>; break;
        }
        if (substream) /* <BLOCK, PROBE, DRAIN, SUB> */
            kern = FB == 1024 ? mega_bvh_kernel<1024, false, false, true> : FB == 768 ? mega_bvh_kernel<768, false, false, true> : FB == 512 ? mega_bvh_kernel<512, false, false, true>
                   : FB == 384 ? mega_bvh_kernel<384, false, false, true> : mega_bvh_kernel<256, false, false, true>;
        /* scheduling thresholds (mega_bvh.h).  Smaller batches do not help a chain-bound partition: measured on
         * one rank of 8, (32,24,16) 100 ms, (12,12,8) 126 ms, (2,2,2) 192 ms -- a lane waits through every step
         * its wave runs for other lanes, and small batches mean more of those */
        {
            fa.th_s = MORT_TH_S; fa.th_l = MORT_TH_L; fa.t_keep = MORT_T_KEEP;
            const char *th = std::getenv("MORT_THRESHOLDS"); /* "s,l,k" */
            if (th) { int s_ = 0, l_ = 0, k_ = 0; if (std::sscanf(th, "%d,%d,%d", &s_, &l_, &k_) == 3) { fa.th_s = s_; fa.th_l = l_; fa.t_keep = k_; } }
        }
        /* LDS: hot blob + as many bounce-stack levels per lane as fit next to it (768 threads: one workgroup per CU) */
        const uint32_t tstack_off = (fa.hot_bytes + 15u) & ~15u; /* traversal stacks: [levels][thread] u16 (four-wide: the world's own bound, at most MORT_OWN4_STACK) */
        const uint32_t stack_off = tstack_off + (uint32_t)c->own4_stack * (uint32_t)FB * 2u; /* as many levels as this world's tree can have pending */
        fa.off_tstack = tstack_off;
        uint32_t static_lds = 1024; /* the kernel's own __shared__ objects come out of the same 160 KB */
        { hipFuncAttributes fattr; if (hipFuncGetAttributes(&fattr, (const void *)kern) == hipSuccess) static_lds = (uint32_t)((fattr.sharedSizeBytes + 1023) & ~(size_t)1023); }
        int groups_per_cu = FB >= 768 ? 1 : 768 / FB; /* keep 12 waves per CU */
        while (groups_per_cu > 1 && (long long)(stack_off + static_lds) * groups_per_cu > 160ll * 1024) groups_per_cu--;
        long long room = (160ll * 1024 - (long long)static_lds * groups_per_cu) / groups_per_cu - (long long)stack_off;
        if (room < 0) { c->last_error = "BVH image does not fit one CU's LDS"; return MORT_ERR_CAPACITY; }
        int dl = (int)(room / ((long long)FB * 16));
        if (dl > 12) dl = 12;
        fa.off_stack = stack_off; fa.stack_lds_depth = dl;
        const size_t lds_bytes = (size_t)stack_off + (size_t)dl * FB * 16;
        HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, FB, lds_bytes) != hipSuccess || per_cu < 1) per_cu = 1;
        { const char *pc = std::getenv("MORT_FAST_BLOCKS_PER_CU"); if (pc && std::atoi(pc) >= 1 && std::atoi(pc) < per_cu) per_cu = std::atoi(pc); }
        int grid = c->num_cus * per_cu;
        const int want_blocks = (int)((lanes_wanted + FB - 1) / FB);
        if (grid > want_blocks) grid = want_blocks;
        if (grid < 1) grid = 1;
        lds_bytes_used = (int)lds_bytes;
        fast_kernel_used = (const void *)kern;
        { /* bounce-stack levels that do not fit in LDS: [level - dl][lane of the launch] in HBM */
            const int deep_levels = cam->bounce_limit > dl ? cam->bounce_limit - dl : 0;
            int st_d = ensure_buf(c, &c->d_deep, &c->deep_cap, (size_t)(deep_levels > 0 ? deep_levels : 1) * (size_t)grid * (size_t)FB * sizeof(float4));
            if (st_d != MORT_OK) return st_d;
            fa.deep = (float4 *)c->d_deep;
            fa.wave_log = nullptr;
            if (std::getenv("MORT_WAVE_LINES")) { /* profile builds: 16 words per wave, read back by mort_hip_debug_wave_log */
                c->wave_log_waves = (size_t)grid * (size_t)(FB / 64);
                if (ensure_buf(c, &c->d_wave_log, &c->wave_log_cap, c->wave_log_waves * 16 * sizeof(unsigned long long)) == MORT_OK) {
                    hipMemsetAsync(c->d_wave_log, 0, c->wave_log_waves * 16 * sizeof(unsigned long long), s);
                    fa.wave_log = (unsigned long long *)c->d_wave_log;
                }
            }
        }
        std::snprintf(kname, sizeof kname, "mega_bvh_kernel<%d, false, %s, %s>", FB, chain_bound ? "true" : "false", substream ? "true" : "false");
        /* ---- tile order: expensive tiles first (sub-stream launches: work items are a stratum row, 1/sqrt_spp of a pixel -- no long tail to order away) ---- */
        if (!substream && !std::getenv("MORT_NO_TILE_ORDER") && tiles >= 4 * grid) {
            int st_o = prepare_tile_order(c, cam, a, fa, tiles, grid, FB, chain_bound, s, [&](const FastArgs &pa) {
                hipError_t e_ = hipFuncSetAttribute((const void *)kern_probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
                if (e_ != hipSuccess) return e_;
                hipLaunchKernelGGL(kern_probe, dim3(grid), dim3(FB), lds_bytes, s, pa);
                return hipGetLastError();
            });
            if (st_o != MORT_OK) return st_o;
            if (stats) HIPCHK(c, hipEventRecord(c->ev0, s)); /* time the frame itself; ordering upkeep is reported by wall-clock benches */
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(FB), lds_bytes, s, fa);
        HIPCHK(c, hipGetLastError());
        if (substream) {
            const int npx = W * a.local_rows;
            hipLaunchKernelGGL(substream_resolve_kernel, dim3((npx + 255) / 256), dim3(256), 0, s, (const float *)c->d_vaccum, W, a.local_rows, cam->sqrt_spp,
                               a.pixel_samples_scale, (uchar4 *)d_rgba, (float *)d_accum);
            HIPCHK(c, hipGetLastError());
        }
    } else if (blocks > 0 && use_gen) {
        GenArgs ga;
        std::memset(&ga, 0, sizeof ga);
        ga = c->gen; /* LDS image offsets, root, far-ray constants (upload_world) */
        FastArgs &fa = ga.f;
        fa.r = a;
        fa.hot_src = (const unsigned char *)c->d_gen; fa.hot_bytes = c->gen_bytes;
        fa.next_q = (unsigned int *)(c->d_counters + 2);
        fa.tiles_x = (W + 7) / 8; fa.tiles_total = tiles;
        if (substream) { fa.sub = cam->sqrt_spp; fa.vaccum = c->d_vaccum; fa.r.states = c->d_substates; }
        const long long lanes_wanted = (long long)tiles * 64;
        int FB = 768;
        bool heavy_default = false;
        { const char *fb_env = std::getenv("MORT_GEN_BLOCK_SIZE");
          if (fb_env && !substream) FB = std::atoi(fb_env);
          else { /* many pixels per lane: 1024 threads compiled for 128 registers = four waves per SIMD, which pays for its spills as in the BVH
                  * kernel (4096 x 4096 x 4: 126.6 ms, 768 threads at 160 registers 137.5 ms, 512 threads 181 ms; 1920 x 1080 x 49: 173 vs 175 ms).
                  * With a handful of pixels per lane the frame ends with its longest pixel chains, whose rounds are faster with two waves per
                  * SIMD and no spills (final scene 800 x 800 x 961: 3.07 s with 512 threads, 3.21 s with 1024); fewer pixels than lanes:
                  * 256-thread groups so every CU has work */
                 const double ppl = (double)lanes_wanted / ((double)c->num_cus * 768.0);
                 FB = ppl >= 8.0 ? 1024 : (lanes_wanted >= 512ll * c->num_cus) ? 512 : 256;
                 /* fewer pixels per lane: the frame may be bound by its longest pixel chains, and then HEAVY WAVES pay (below): 1024 threads, the launch decides on the device */
                 if (ppl < 8.0 && lanes_wanted >= 32ll * 1024 && !std::getenv("MORT_GEN_NO_HEAVY")) { FB = 1024; heavy_default = true; } } }
        if (FB != 1024 && FB != 768 && FB != 512 && FB != 256) FB = 256;
        if (FB == 1024 && (size_t)c->gen_bytes + 16u + (size_t)MORT_OWN_STACK * 1024u * 2u + 2048u > 160u * 1024u) FB = 768; /* image + traversal stacks of 1024 threads must fit */
        if (substream && FB > 512) FB = 512; /* the non-parity launch is instantiated for 512- and 256-thread workgroups */
        /* swept on the final scene, 800x800x100 (scripts/th_sweep.py, 180 settings): 373 ms here vs 449 ms with the BVH kernel's (40,24,12) and m = 24 */
        fa.th_s = 28; fa.th_l = 20; fa.t_keep = 4; ga.th_m = 56;
        fa.lane_cap = 64;
        { const char *lc = std::getenv("MORT_LANE_CAP"); if (lc && std::atoi(lc) >= 1 && std::atoi(lc) <= 64) fa.lane_cap = std::atoi(lc); }
        /* heavy waves (mega_bvh.h FastArgs.heavy_*): "mod,num,cap,percent" */
        fa.heavy_mod = 0; fa.heavy_num = 0; fa.heavy_cap = 64; c->heavy_percent = 50;
        /* default where the frame has fewer than 8 pixels per lane: two waves of three take 12 lanes each from the tiles whose longest pixel reaches half of the frame's
         * longest -- IF the device finds the frame chain-bound (tile_sort.hip heavy_count_kernel); final scene 800x800x961: 3.04 -> 2.55 s, x100: 318 -> 270 ms */
        if (heavy_default && FB == 1024 && !substream) { fa.heavy_mod = 3; fa.heavy_num = 2; fa.heavy_cap = 12; }
        { const char *hv = std::getenv("MORT_GEN_HEAVY"); int m_ = 0, n_ = 0, k_ = 0, p_ = 0;
          if (hv && !substream && std::sscanf(hv, "%d,%d,%d,%d", &m_, &n_, &k_, &p_) == 4 && m_ >= 1 && n_ >= 1 && n_ <= m_ && k_ >= 1 && k_ <= 64 && p_ >= 1 && p_ <= 100) {
              fa.heavy_mod = m_; fa.heavy_num = n_; fa.heavy_cap = k_; c->heavy_percent = p_; } }
        ga.drain_mode = 3; /* thresholds as shares of the live lanes; measured alternatives: 0 = fixed counts, 1 = follow one lane, 2 = rounds */
        { const char *dm = std::getenv("MORT_GEN_DRAIN"); if (dm) ga.drain_mode = std::atoi(dm); }
        { const char *th = std::getenv("MORT_GEN_THRESHOLDS"); /* "s,l,k,m" */
          if (th) { int s_ = 0, l_ = 0, k_ = 0, m_ = 0; if (std::sscanf(th, "%d,%d,%d,%d", &s_, &l_, &k_, &m_) == 4) { fa.th_s = s_; fa.th_l = l_; fa.t_keep = k_; ga.th_m = m_; } } }
        const uint32_t tstack_off = (c->gen_bytes + 15u) & ~15u;
        const uint32_t stack_off = tstack_off + (uint32_t)MORT_OWN_STACK * (uint32_t)FB * 2u;
        fa.off_tstack = tstack_off;
        uint32_t static_lds = 1024; /* the kernel's own __shared__ objects come out of the same 160 KB */
        { hipFuncAttributes fattr; if (mort_gen_attributes(FB, ga.prims_in_lds != 0, &fattr, substream) == hipSuccess) static_lds = (uint32_t)((fattr.sharedSizeBytes + 1023) & ~(size_t)1023); }
        /* workgroups per CU: as many as keep 12 waves per CU, but a big image (the final scene's 120 KB) admits one */
        int groups_per_cu = (FB == 512 || FB == 1024) ? 1 : 768 / FB;
        while (groups_per_cu > 1 && (long long)(stack_off + static_lds) * groups_per_cu > 160ll * 1024) groups_per_cu--;
        long long room = (160ll * 1024 - (long long)static_lds * groups_per_cu) / groups_per_cu - (long long)stack_off;
        if (room < 0) { c->last_error = "unified-tree image does not fit one CU's LDS"; return MORT_ERR_CAPACITY; }
        int dl = (int)(room / ((long long)FB * 16));
        if (dl > 12) dl = 12;
        { const char *de = std::getenv("MORT_GEN_DL"); if (de && std::atoi(de) < dl) dl = std::atoi(de); }
        fa.off_stack = stack_off; fa.stack_lds_depth = dl;
        const size_t lds_bytes = (size_t)stack_off + (size_t)dl * FB * 16;
        int per_cu = mort_gen_blocks_per_cu(FB, ga.prims_in_lds != 0, lds_bytes, substream);
        if (per_cu < 1) per_cu = 1;
        int grid = c->num_cus * per_cu;
        const int want_blocks = (int)((lanes_wanted + FB - 1) / FB);
        if (grid > want_blocks) grid = want_blocks;
        if (grid < 1) grid = 1;
        lds_bytes_used = (int)lds_bytes;
        gen_block_used = FB;
        { /* bounce-stack levels that do not fit in LDS: [level - dl][lane of the launch] in HBM */
            const int deep_levels = cam->bounce_limit > dl ? cam->bounce_limit - dl : 0;
            int st_d = ensure_buf(c, &c->d_deep, &c->deep_cap, (size_t)(deep_levels > 0 ? deep_levels : 1) * (size_t)grid * (size_t)FB * sizeof(float4));
            if (st_d != MORT_OK) return st_d;
            fa.deep = (float4 *)c->d_deep;
            fa.wave_log = nullptr;
            if (std::getenv("MORT_WAVE_LINES")) { /* profile builds: 16 words per wave, read back by mort_hip_debug_wave_log */
                c->wave_log_waves = (size_t)grid * (size_t)(FB / 64);
                if (ensure_buf(c, &c->d_wave_log, &c->wave_log_cap, c->wave_log_waves * 16 * sizeof(unsigned long long)) == MORT_OK) {
                    hipMemsetAsync(c->d_wave_log, 0, c->wave_log_waves * 16 * sizeof(unsigned long long), s);
                    fa.wave_log = (unsigned long long *)c->d_wave_log;
                }
            }
        }
        std::snprintf(kname, sizeof kname, substream ? "mega_gen_kernel<%d, %s, true>" : "mega_gen_kernel<%d, %s, false>", FB, ga.prims_in_lds ? "true" : "false"); /* as rocprofv3 prints it: <BLOCK, PRIMS_LDS, SUB> */
        if (!substream && !std::getenv("MORT_NO_TILE_ORDER") && tiles >= 4 * grid) {
            int st_o = prepare_tile_order(c, cam, a, fa, tiles, grid, FB, false, s, [&](const FastArgs &pa) {
                GenArgs pg = ga;
                pg.f = pa;
                pg.probe = 1;
                return mort_gen_launch(pg, FB, grid, lds_bytes, s);
            });
            if (st_o != MORT_OK) return st_o;
            if (stats) HIPCHK(c, hipEventRecord(c->ev0, s));
            /* priority pixels (mega_bvh.h FastArgs): the head of the cost order, a few per wave.  A frame with a handful of pixels per lane
             * ends when its longest pixel chain does (final scene 800x800: the fog ball's pixels run 8 x the mean), and a chain advances one
             * segment per round of its wave: such a pixel must not share its wave with 63 others of its kind, and its wave must follow it */
            int k_prio = 0; /* measured, not the default: following one lane starves the other 63 of a tile whose pixels are all long */
            { const char *kp = std::getenv("MORT_GEN_PRIO_LANES"); if (kp) k_prio = std::atoi(kp); }
            if (k_prio > 0 && ga.drain_mode == 1 && !substream) {
                const long long waves = (long long)grid * (FB / 64);
                long long pt = (waves * k_prio + 63) / 64;
                if (pt > tiles / 4) pt = tiles / 4;
                fa.prio_tiles = (int)pt; fa.prio_lanes = k_prio;
            }
        }
        HIPCHK(c, mort_gen_launch(ga, FB, grid, lds_bytes, s));
        if (fa.heavy_mod > 0 && c->d_prio_count && std::getenv("MORT_GEN_HEAVY_DEBUG")) { /* diagnostic: what the device decided for this launch */
            unsigned h[4] = {0, 0, 0, 0};
            HIPCHK(c, hipStreamSynchronize(s));
            HIPCHK(c, hipMemcpy(h, c->d_prio_count, 16, hipMemcpyDeviceToHost));
            std::fprintf(stderr, "[heavy] head tiles %u of %d, lanes %d\n", h[0], tiles, grid * FB);
        }
        if (fa.heavy_mod > 0 && c->d_prio_count) /* this frame's segment total, beside the tile costs it leaves for the next frame's order */
            HIPCHK(c, hipMemcpyAsync(c->d_prio_count + 4, c->d_counters, 96 * sizeof(unsigned long long), hipMemcpyDeviceToDevice, s));
        if (substream) {
            const int npx = W * a.local_rows;
            hipLaunchKernelGGL(substream_resolve_kernel, dim3((npx + 255) / 256), dim3(256), 0, s, (const float *)c->d_vaccum, W, a.local_rows, cam->sqrt_spp,
                               a.pixel_samples_scale, (uchar4 *)d_rgba, (float *)d_accum);
            HIPCHK(c, hipGetLastError());
        }
    } else if (blocks > 0) {
        hipLaunchKernelGGL(mega_kernel, dim3(blocks), dim3(64 * waves_per_block), 0, s, a);
        HIPCHK(c, hipGetLastError());
    }
    if (stats) {
        HIPCHK(c, hipEventRecord(c->ev1, s));
        /* what the statistics need, by value: mort_hip_render_gather lets the frame gather follow the render on the stream
         * and collects them after its one wait (c->defer_stats) */
        const int sqrt_spp_ = cam->sqrt_spp, local_rows_ = a.local_rows;
        const bool has_accum_ = d_accum != nullptr;
        const std::string kname_ = kname;
        auto fill = [=](mort_stats *stats) -> int {
        HIPCHK(c, hipEventSynchronize(c->ev1));
        float ms = 0;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
        unsigned long long cnt[96] = {0};
        HIPCHK(c, hipMemcpy(cnt, c->d_counters, sizeof cnt, hipMemcpyDeviceToHost));
#ifdef MORT_PROFILE_STATES
        if (mode == MORT_MODE_WAVE) {
            const char *nm[3] = {"T", "L", "F"};
            const double tot = (double)(cnt[12] + cnt[13] + cnt[14] + cnt[15]);
            for (int k = 0; k < 3; k++)
                std::fprintf(stderr, "[wf_trav %s] %10llu wave-steps  util %5.1f%%  cycles %5.1f%% (%.0f/step)\n", nm[k], cnt[4 + k],
                             cnt[4 + k] ? 100.0 * (double)cnt[8 + k] / (64.0 * (double)cnt[4 + k]) : 0.0, 100.0 * (double)cnt[12 + k] / tot,
                             cnt[4 + k] ? (double)cnt[12 + k] / (double)cnt[4 + k] : 0.0);
            std::fprintf(stderr, "[wf_trav sched] cycles %5.1f%%   fronts %d\n", 100.0 * (double)cnt[15] / tot, c->wf_fronts);
            std::fprintf(stderr, "[wf_trav waves] %llu waves, mean lifetime %.1f us, in-loop cycles per wave %.0f\n", cnt[21],
                         cnt[21] ? (double)cnt[20] / (double)cnt[21] * 0.01 : 0.0, cnt[21] ? tot / (double)cnt[21] : 0.0);
        } else if (use_gen) {
            const char *nm[4] = {"T", "L", "M", "S"};
            unsigned long long segs = cnt[0];
            for (int k = 0; k < 32; k++) segs += cnt[32 + 2 * k];
            const double tot = (double)(cnt[12] + cnt[13] + cnt[14] + cnt[15] + cnt[16]);
            for (int k = 0; k < 4; k++)
                std::fprintf(stderr, "[gen %s] %12llu wave-steps  lanes %5.1f%%  cycles %5.1f%% (%.0f/step)\n", nm[k], cnt[4 + 2 * k],
                             cnt[4 + 2 * k] ? 100.0 * (double)cnt[5 + 2 * k] / (64.0 * (double)cnt[4 + 2 * k]) : 0.0, 100.0 * (double)cnt[12 + k] / tot,
                             cnt[4 + 2 * k] ? (double)cnt[12 + k] / (double)cnt[4 + 2 * k] : 0.0);
            std::fprintf(stderr, "[gen S parts, cycles per S step] scan+decode %.0f  shade call %.0f  stack store %.0f  finish %.0f  newpix %.0f  newray %.0f\n", (double)cnt[24] / (double)cnt[10],
                         (double)cnt[25] / (double)cnt[10], (double)cnt[20] / (double)cnt[10], (double)cnt[21] / (double)cnt[10], (double)cnt[22] / (double)cnt[10], (double)cnt[23] / (double)cnt[10]);
            std::fprintf(stderr, "[gen sched] cycles %5.1f%%; leaf loop: %.2f iterations per L step, %.1f lanes per iteration; scans %llu; steps per segment: T %.2f L %.2f M %.2f S %.2f\n",
                         100.0 * (double)cnt[16] / tot, cnt[6] ? (double)cnt[17] / (double)cnt[6] : 0.0, cnt[17] ? (double)cnt[18] / (double)cnt[17] : 0.0, cnt[3],
                         (double)cnt[5] / (double)(segs + 1), (double)cnt[7] / (double)(segs + 1), (double)cnt[9] / (double)(segs + 1), (double)cnt[11] / (double)(segs + 1));
        } else if (use_fast) {
            const char *nm[3] = {"T", "L", "S"};
            for (int k = 0; k < 3; k++)
                std::fprintf(stderr, "[states] %s: %llu wave-steps, %llu lane-steps, utilisation %.1f%%\n", nm[k], cnt[4 + 2 * k], cnt[5 + 2 * k],
                             cnt[4 + 2 * k] ? 100.0 * (double)cnt[5 + 2 * k] / (64.0 * (double)cnt[4 + 2 * k]) : 0.0);
            std::fprintf(stderr, "[states] box-step runs: %llu (%.1f steps per run)\n", cnt[30], cnt[30] ? (double)cnt[4] / (double)cnt[30] : 0.0);
            const double tot = (double)(cnt[10] + cnt[11] + cnt[12] + cnt[13]);
            std::fprintf(stderr, "[cycles] T %.1f%% (%.0f/step)  L %.1f%% (%.0f/step)  S %.1f%% (%.0f/step)  sched %.1f%%  total wave-cycles %.3g\n",
                         100.0 * cnt[10] / tot, (double)cnt[10] / (double)cnt[4], 100.0 * cnt[11] / tot, (double)cnt[11] / (double)cnt[6],
                         100.0 * cnt[12] / tot, (double)cnt[12] / (double)cnt[8], 100.0 * cnt[13] / tot, tot);
#ifdef MORT_PROFILE_FINE
            std::fprintf(stderr, "[S shade parts, cycles/step] verify %.0f  hit record %.0f  metal %.0f  dielectric %.0f  lambert texture %.0f  lambert scatter %.0f  light %.0f  (rest of 'shade' below: stack store)\n",
                         (double)cnt[18] / (double)cnt[8], (double)cnt[19] / (double)cnt[8], (double)cnt[20] / (double)cnt[8], (double)cnt[21] / (double)cnt[8],
                         (double)cnt[22] / (double)cnt[8], (double)cnt[23] / (double)cnt[8], (double)cnt[24] / (double)cnt[8]);
#else
            {
                const char *bn[6] = {"metal", "dielectric", "lambertian", "finish", "get_ray", "unwind iteration"};
                for (int k = 0; k < 6; k++)
                    std::fprintf(stderr, "[S branch] %-16s entered in %5.1f%% of S steps (x%.2f), %4.1f lanes when entered\n", bn[k],
                                 100.0 * (double)cnt[18 + 2 * k] / (double)cnt[8], (double)cnt[18 + 2 * k] / (double)cnt[8],
                                 cnt[18 + 2 * k] ? (double)cnt[19 + 2 * k] / (double)cnt[18 + 2 * k] : 0.0);
            }
#endif
            std::fprintf(stderr, "[S parts, cycles/step] shade %.0f  finish %.0f  newpix %.0f  newsample+setup %.0f\n", (double)cnt[14] / (double)cnt[8],
                         (double)cnt[15] / (double)cnt[8], (double)cnt[16] / (double)cnt[8], (double)cnt[17] / (double)cnt[8]);
        }
#endif
        std::memset(stats, 0, sizeof *stats);
        stats->seconds = ms * 1e-3;
        for (int k = 0; k < 32; k++) { cnt[0] += cnt[32 + 2 * k]; cnt[1] += cnt[33 + 2 * k]; } /* BVH megakernel: per-workgroup slots */
        stats->segments = cnt[0];
        stats->rng_draws = cnt[1];
        stats->reference_walks = ((use_fast || use_gen) && mode != MORT_MODE_WAVE) ? cnt[3] : 0;
        stats->pixels = (uint64_t)W * (uint64_t)local_rows_;
        stats->eff_samples = stats->pixels * (uint64_t)(sqrt_spp_ * sqrt_spp_);
        stats->algorithmic_hbm_bytes = stats->pixels * (uint64_t)(100 + (has_accum_ ? 12 : 0));
        stats->scene_in_lds = (use_fast || use_gen || mode == MORT_MODE_WAVE) ? 1 : 0;
        if (mode == MORT_MODE_WAVE) stats->algorithmic_hbm_bytes += 240ull * stats->segments; /* wave_bvh.h: per-segment record traffic */
        stats->local_rows = local_rows_;
        std::snprintf(stats->kernel_name, sizeof stats->kernel_name, "%s", kname_.c_str());
        hipFuncAttributes fattr;
        const void *kf = (mode == MORT_MODE_WAVE && wave_gen) ? mort_wave_gen_trav_kernel(c->gen.prims_in_lds != 0, nullptr)
                         : mode == MORT_MODE_WAVE ? (const void *)wf_trav<MORT_WF_BLOCK> : !use_fast ? (const void *)mega_kernel
                         : fast_kernel_used;
        const bool gen_ran = use_gen && mode != MORT_MODE_WAVE && gen_block_used > 0;
        if ((gen_ran ? mort_gen_attributes(gen_block_used, c->gen.prims_in_lds != 0, &fattr, substream) : hipFuncGetAttributes(&fattr, kf)) == hipSuccess) {
            stats->kernel_vgprs = fattr.numRegs;
            stats->kernel_lds_bytes = (use_fast || gen_ran) ? lds_bytes_used : (int)fattr.sharedSizeBytes;
        }
            return MORT_OK;
        };
        if (c->defer_stats) { c->pending_stats = fill; return MORT_OK; }
        return fill(stats);
    }
    return MORT_OK;
}

/* diagnostic (not in include/mort_hip.h): the per-wave records of the last frame of a profile build run with MORT_WAVE_LINES=1 */
extern "C" int mort_hip_debug_wave_log(mort_ctx *c, unsigned long long *out, size_t max_waves) {
    if (!c || !out || !c->d_wave_log) return 0;
    hipSetDevice(c->device);
    quiesce(c);
    const size_t n = c->wave_log_waves < max_waves ? c->wave_log_waves : max_waves;
    if (hipMemcpy(out, c->d_wave_log, n * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return (int)n;
}

/* diagnostic (not in include/mort_hip.h; host only, no HIP call): facts of this build's own trees over a world -- out[0] binary
 * nodes, [1] leaves, [2] binary depth, [3] four-wide nodes, [4] its pending-children bound, [5] leaf references reached from the
 * four-wide root, [6] leaves reached more than once or out of range, [7] child slots in use, [8] 1 if every four-wide box and margin
 * is bit for bit one of the binary tree's for the same child. */
extern "C" int mort_hip_debug_own_tree(const mort_world *w, int *out) {
    if (!w || !out) return MORT_ERR_INVALID;
    SceneBlob sb;
    const int st = build_scene_blob(w, sb);
    if (st != MORT_OK) return st;
    const mortc::Compiled &o = sb.comp;
    for (int i = 0; i < 9; i++) out[i] = 0;
    out[0] = (int)o.own_nodes.size(); out[1] = (int)o.own_leaves.size(); out[2] = o.own_depth;
    out[3] = (int)o.own_nodes4.size(); out[4] = o.own4_stack;
    if (o.own_nodes4.empty()) return MORT_OK;
    /* child reference -> (box, margin) in the binary tree */
    struct Rec { float v[7]; };
    std::vector<Rec> of_inner(o.own_nodes.size()), of_leaf(o.own_leaves.size());
    for (const DNode2 &nd : o.own_nodes)
        for (int k = 0; k < 2; k++) {
            const uint32_t ref = k ? nd.child1 : nd.child0;
            Rec r = k ? Rec{{nd.x1min, nd.x1max, nd.y1min, nd.y1max, nd.z1min, nd.z1max, nd.e1}} : Rec{{nd.x0min, nd.x0max, nd.y0min, nd.y0max, nd.z0min, nd.z0max, nd.e0}};
            if (ref & 0x8000u) of_leaf[ref & 0x7fffu] = r; else of_inner[ref] = r;
        }
    std::vector<int> seen(o.own_leaves.size(), 0);
    std::vector<uint32_t> todo(1, 0u);
    bool same = true;
    /* every inner child of a four-wide node stands for one binary node: find which by its box, via the leaves below it */
    while (!todo.empty()) {
        const uint32_t n = todo.back(); todo.pop_back();
        const DNode4 &nd = o.own_nodes4[n];
        for (int k = 0; k < 4; k++) {
            const uint32_t ref = nd.child[k];
            if (ref == 0xffffu) continue;
            out[7]++;
            const Rec r{{nd.xmin[k], nd.xmax[k], nd.ymin[k], nd.ymax[k], nd.zmin[k], nd.zmax[k], nd.e[k]}};
            if (ref & 0x8000u) {
                const uint32_t l = (ref & 0x7fffu) / MORT_LEAF2_PIECES;
                if ((ref & 0x7fffu) % MORT_LEAF2_PIECES || l >= seen.size() || seen[l]++) { out[6]++; continue; }
                out[5]++;
                if (std::memcmp(&r, &of_leaf[l], sizeof r) != 0) same = false;
            } else {
                if (ref % MORT_NODE4_PIECES || ref / MORT_NODE4_PIECES >= o.own_nodes4.size()) { out[6]++; continue; }
                bool found = false;
                for (const Rec &b : of_inner) if (std::memcmp(&r, &b, sizeof r) == 0) { found = true; break; }
                if (!found) same = false;
                todo.push_back(ref / MORT_NODE4_PIECES);
            }
        }
    }
    out[8] = same ? 1 : 0;
    return MORT_OK;
}

extern "C" int mort_hip_render_device(mort_ctx *c, const mort_camera *cam, int mode, void *d_rgba, void *d_accum,
                                      void *stream, mort_stats *stats) {
    return render_device_impl(c, cam, mode, d_rgba, d_accum, nullptr, stream, stats);
}

static int ensure_buf(mort_ctx *c, void **p, size_t *cap, size_t need) {
    if (*cap >= need && *p) return MORT_OK;
    if (*p) { hipFree(*p); *p = nullptr; *cap = 0; }
    HIPCHK(c, hipMalloc(p, need ? need : 16));
    *cap = need;
    return MORT_OK;
}

extern "C" int mort_hip_render(mort_ctx *c, const mort_camera *cam, int mode, uint8_t *rgba_out, float *accum_out,
                               uint32_t *segments_px_out, mort_stats *stats) {
    if (!c || !cam || !rgba_out) return MORT_ERR_INVALID;
    const int W = cam->image_width, H = cam->image_height;
    if (W <= 0 || H <= 0) return MORT_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    const int lr = local_rows_for(c->part, H);
    const size_t npx = (size_t)W * (size_t)lr;
    int st;
    if ((st = ensure_buf(c, &c->d_rgba, &c->rgba_cap, npx * 4)) != MORT_OK) return st;
    if (accum_out && (st = ensure_buf(c, &c->d_accum, &c->accum_cap, npx * 12)) != MORT_OK) return st;
    if (segments_px_out) { if ((st = ensure_buf(c, &c->d_segpx, &c->segpx_cap, npx * 4)) != MORT_OK) return st; }
    else if (c->d_segpx) { hipFree(c->d_segpx); c->d_segpx = nullptr; c->segpx_cap = 0; }
    mort_stats local;
    st = render_device_impl(c, cam, mode, c->d_rgba, accum_out ? c->d_accum : nullptr,
                            segments_px_out ? (uint32_t *)c->d_segpx : nullptr, nullptr, &local);
    if (st != MORT_OK) return st;
    if (stats) *stats = local;
    if (c->part.nranks == 1) { /* packed rows are the whole image: three copies instead of one per row */
        HIPCHK(c, hipMemcpy(rgba_out, c->d_rgba, npx * 4, hipMemcpyDeviceToHost));
        if (accum_out) HIPCHK(c, hipMemcpy(accum_out, c->d_accum, npx * 12, hipMemcpyDeviceToHost));
        if (segments_px_out) HIPCHK(c, hipMemcpy(segments_px_out, c->d_segpx, npx * 4, hipMemcpyDeviceToHost));
        return MORT_OK;
    }
    for (int ly = 0; ly < lr; ly++) {
        const int y = global_row_host(c->part, ly);
        HIPCHK(c, hipMemcpy(rgba_out + (size_t)y * W * 4, (uint8_t *)c->d_rgba + (size_t)ly * W * 4, (size_t)W * 4, hipMemcpyDeviceToHost));
        if (accum_out) HIPCHK(c, hipMemcpy(accum_out + (size_t)y * W * 3, (float *)c->d_accum + (size_t)ly * W * 3, (size_t)W * 12, hipMemcpyDeviceToHost));
        if (segments_px_out) HIPCHK(c, hipMemcpy(segments_px_out + (size_t)y * W, (uint32_t *)c->d_segpx + (size_t)ly * W, (size_t)W * 4, hipMemcpyDeviceToHost));
    }
    return MORT_OK;
}
