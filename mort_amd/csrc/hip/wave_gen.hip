/*
 * wave_gen.hip -- MORT_MODE_WAVE for worlds without reference BVHs (reference scenes 2..9; BASELINE config 5 is the
 * book-2 final scene through this form): the render path as a queue of path segments instead of one lane per pixel.
 *
 * One path per pixel, all pixels of the rank in flight.  A pixel's samples stay chained through its own XORWOW state
 * (the reference's streams are pixel-serial, rng.cuh:17-23), so a FRONT is one segment of every live pixel: a compact
 * array of (path id, ray) records in HBM.  Per front, two launches:
 *
 *   wf_trav_gen   closest solid hit of every record: the unified tree (scene_compile.h build_unified) in LDS, walked by
 *                 the same per-lane state machine as mega_gen_kernel (T both child boxes of a node / L the primitives of
 *                 a leaf / F retire + refill).  Each wave owns 64-record batches of the front, strided over the front; a
 *                 lane whose ray is done takes the next record by rank among the waiting lanes (ballot + mbcnt), so the
 *                 box-test loop stays full.  Retiring writes an 8-byte hit record at the ray's position and appends the
 *                 position to the queue of its shade class, staged per wave in LDS (one global atomic per ~128 entries).
 *   wf_shade_gen  one workgroup = 256 positions of ONE class (diffuse / specular+emissive / miss): the constant media
 *                 after the solids (they draw from the pixel's stream, which lives here: constant_medium::hit,
 *                 objects.cuh:396-434), then dev_shade.h's shade_hit -- every material, texture and the light mixture --
 *                 then either the next segment's record (appended to the next front: one atomic per workgroup,
 *                 coalesced 32-byte writes) or the unwind, the next sample's camera ray, or the finished pixel.
 *
 * Same streams, same operations, same order per pixel as the megakernels: bit-identical output (tests/test_gpu_gen.py).
 * Algorithmic HBM traffic per segment: trav 32 r + 8 w + 4 q; shade 4 q + 4 + 32 + 8 + 48 r, 4 + 32 + 48 + 16 w = 240 B.
 */
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "wave_gen.h"
#include "wave_common.h"
#include "dev_gen.h"
#include "dev_shade_call.h"

#pragma clang fp contract(off)

struct WfGenArgs {
    GenArgs g;                /* g.f.r: camera / partition / buffers; g.f.hot_src / hot_bytes / off_tstack: LDS image */
    uint32_t t_stage;         /* LDS offset of the per-wave class staging */
    int n_paths;
    unsigned *q_id[2];        /* front[parity]: path id of each record */
    WfRay *q_ray[2];          /* front[parity]: its ray */
    WfHit *hits;              /* by position in the current front */
    WfPix *pix;               /* by path id */
    float4 *stack;            /* [bounce_limit][n_paths] */
    unsigned *q_cls[3];       /* positions, per shade class */
    WfCounters *cnt;
    int parity;               /* front & 1 */
};

#define WG_STAGE 128 /* positions per class staged in LDS per wave before a flush */
enum { W_T = 0, W_L = 1, W_F = 2, W_DONE = 3 };

/* ---- front 0: load streams, first camera ray of every pixel ---- */
__global__ void __launch_bounds__(256) wf_init_gen(const WfGenArgs w) {
    const RenderArgs &a = w.g.f.r;
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id == 0) {
        w.cnt->front_count[0] = (unsigned)w.n_paths; w.cnt->front_count[1] = 0;
        for (int p = 0; p < 2; p++) for (int k = 0; k < 3; k++) w.cnt->cls_count[p][k] = 0;
        w.cnt->live = (unsigned)w.n_paths;
    }
    if (id >= w.n_paths) return;
    const int ly = id / a.width, x = id - ly * a.width;
    const int y = global_row(ly, a.rank, a.nranks, a.rows_per_block);
    const mort_rng_state st = a.states[id];
    Rng rng; rng.d = st.d; rng.v0 = st.v[0]; rng.v1 = st.v[1]; rng.v2 = st.v[2]; rng.v3 = st.v[3]; rng.v4 = st.v[4]; rng.draws = 0;
    const Ray ray = get_ray(a, x, y, rng, 0, 0);
    WfRay rr; rr.ox = ray.o.x; rr.oy = ray.o.y; rr.oz = ray.o.z; rr.tm = ray.tm; rr.dx = ray.d.x; rr.dy = ray.d.y; rr.dz = ray.d.z; rr.time0 = ray.tm;
    w.q_ray[0][id] = rr;
    w.q_id[0][id] = (unsigned)id;
    WfPix p; wf_rng_store(p, rng); p.cr = p.cg = p.cb = 0; p.packed = 0; p.segments = 1; /* the segment this ray is about to trace */
    w.pix[id] = p;
}

/* the scan for rays the walk does not decide: out of line, HBM copy of the scene */
struct WScanHit { uint32_t best; float closest; };
__device__ __attribute__((noinline)) WScanHit wf_scan_solids(const DScene *scp, int first_medium, const int *chains, int n_chains,
                                                             float ox, float oy, float oz, float dx, float dy, float dz, float tm) {
    Ray r; r.o = mk(ox, oy, oz); r.d = mk(dx, dy, dz); r.tm = tm;
    WScanHit h;
    gen_scan_solids(*scp, first_medium, chains, n_chains, r, h.closest, h.best);
    return h;
}

template <int BLOCK, bool PRIMS_LDS>
__global__ void __launch_bounds__(BLOCK) wf_trav_gen(const WfGenArgs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    /* launch arguments without private memory (mega_gen.hip): the by-value parameter is never named; the kernarg segment is copied into
     * LDS, the loop's wave-uniform values are pinned in SGPRs, the out-of-line scan gets the segment's own address */
    __shared__ WfGenArgs s_w;
    const WfGenArgs *const wp = (const WfGenArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    {
        const uint32_t *src = (const uint32_t *)wp;
        uint32_t *dst = (uint32_t *)&s_w;
        for (uint32_t i = threadIdx.x; i < (uint32_t)(sizeof(WfGenArgs) / 4); i += BLOCK) dst[i] = src[i];
    }
    {
        const uint4 *src = (const uint4 *)wp->g.f.hot_src;
        uint4 *dst = (uint4 *)lds;
        const uint32_t n16 = wp->g.f.hot_bytes >> 4;
        for (uint32_t i = threadIdx.x; i < n16; i += BLOCK) dst[i] = src[i];
    }
    const int par = wp->parity;
    WfCounters *const wcnt = wp->cnt;
    if (blockIdx.x == 0 && threadIdx.x == 0) wcnt->front_count[par ^ 1] = 0; /* next front: its last reader was the previous wf_trav_gen */
    __syncthreads();
    const WfGenArgs &W = s_w;
    const GenArgs &L = W.g;
    const DNodeQ *nodes2 = (const DNodeQ *)(lds + uni_u(L.o_nodes));
    const uint32_t *entries = (const uint32_t *)(lds + uni_u(L.o_entries));
    const int *chains = (const int *)(lds + uni_u(L.o_chains));
    unsigned short *tstack = (unsigned short *)(lds + uni_u(L.f.off_tstack)) + threadIdx.x; /* [level * BLOCK] */
    unsigned *stage = (unsigned *)(lds + uni_u(W.t_stage)) + (threadIdx.x >> 6) * (3 * WG_STAGE);
    DScene lsc; /* only what the leaf test reads: transform chains and primitives */
    lsc.xforms = (const DXform *)(lds + uni_u(L.o_xforms));
    if (PRIMS_LDS) { lsc.spheres = (const DSphere *)(lds + uni_u(L.o_spheres)); lsc.quads = (const DQuad *)(lds + uni_u(L.o_quads)); }
    else { lsc.spheres = uni_p(L.f.r.sc.spheres); lsc.quads = uni_p(L.f.r.sc.quads); }
    const DSphere *spheres = lsc.spheres;
    const DQuad *quads = lsc.quads;
    const uint32_t *ranks = uni_p(L.ranks);
    const int n_spheres = uni_i(L.n_spheres);
    const uint32_t root = uni_u(L.root);
    const float g_x = uni_f(L.gx), g_y = uni_f(L.gy), g_z = uni_f(L.gz), g_R = uni_f(L.gR), g_mnear = uni_f(L.mnear), g_kmin = uni_f(L.kmin);
    WfHit *const hits = uni_p(W.hits);
    unsigned *const q_cls0 = uni_p(W.q_cls[0]), *const q_cls1 = uni_p(W.q_cls[1]), *const q_cls2 = uni_p(W.q_cls[2]);
    int staged[3] = {0, 0, 0};
    const int lane = threadIdx.x & 63;
    const unsigned n_items = wcnt->front_count[par];
    const WfRay *front = par ? uni_p(W.q_ray[1]) : uni_p(W.q_ray[0]);

    /* this wave's share of the front: 64-record batches wave_id, wave_id + n_waves, ... (strided, so every wave samples
     * the whole image: fronts are in pixel order and cost varies by region) */
    const unsigned n_waves = gridDim.x * (BLOCK / 64), wave_id = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    unsigned next_batch = wave_id, b_base = 0;
    int b_cnt = 0, b_off = 0;

    const int th_f = uni_i(L.f.th_s), th_l = uni_i(L.f.th_l), t_keep = uni_i(L.f.t_keep); /* scheduling thresholds (lanes), as in mega_gen.hip; F plays S's part */
    int state = W_F;
    bool have = false;
    unsigned pos = 0;
    Ray ray; ray.o = mk(0, 0, 0); ray.d = mk(0, 0, 1); ray.tm = 0;
    GenRay gr; gr.ix = gr.iy = gr.iz = 1; gr.mx = gr.my = gr.mz = 0; gr.band = 0;
    float ray_a = 1, closest = 0;
    uint32_t best = GBEST_NONE, node = 0;
    int sp = 0, flags = 0;

#define WG_FLUSH(k) do { \
        if (staged[k] > 0) { \
            unsigned base_ = 0; \
            if (lane == 0) base_ = atomicAdd(&wcnt->cls_count[par][k], (unsigned)staged[k]); \
            base_ = __shfl(base_, 0); \
            unsigned *const qc_ = (k) == 0 ? q_cls0 : (k) == 1 ? q_cls1 : q_cls2; \
            for (int i_ = lane; i_ < staged[k]; i_ += 64) qc_[base_ + (unsigned)i_] = stage[(k) * WG_STAGE + i_]; \
            staged[k] = 0; \
        } } while (0)

#ifdef MORT_PROFILE_STATES /* per-state wave-steps, lanes and cycles (scripts/wave_profile.py; printed by mort_hip_render) */
    unsigned long long pr_steps[3] = {0, 0, 0}, pr_lanes[3] = {0, 0, 0}, pr_cyc[4] = {0, 0, 0, 0};
    unsigned long long pt0 = __builtin_readcyclecounter(), pt1;
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#define WGPROF(i, lanes) do { pr_steps[i] += 1; pr_lanes[i] += (unsigned long long)(lanes); } while (0)
#define WGPROFC(i) do { pt1 = __builtin_readcyclecounter(); pr_cyc[i] += pt1 - pt0; pt0 = pt1; } while (0)
#else
#define WGPROF(i, lanes) do { } while (0)
#define WGPROFC(i) do { } while (0)
#endif
    for (;;) {
        const int nT = __popcll(__ballot(state == W_T));
        const int nL = __popcll(__ballot(state == W_L));
        const int nF = __popcll(__ballot(state == W_F));
        if (nT + nL + nF == 0) break;
        int pick;
        if (nF >= th_f) pick = W_F;
        else if (nL >= th_l) pick = W_L;
        else if (nT > 0) pick = W_T;
        else pick = (nL >= nF) ? W_L : W_F;
        WGPROFC(3);

        if (pick == W_T) {
            int keep;
            do {
                WGPROF(0, __popcll(__ballot(state == W_T)));
                if (state == W_T) { /* both child boxes of one node (dev_gen.h gen_prune) */
                    const uint4 *np = (const uint4 *)(nodes2 + node); /* two ds_read_b128: corner + steps, 12 plane offsets + children */
                    const uint4 na = np[0], nb = np[1];
                    const GenBoxes nd = gen_node_decode(__uint_as_float(na.x), __uint_as_float(na.y), __uint_as_float(na.z), na.w, nb.x, nb.y, nb.z, nb.w);
                    float te0, te1;
                    const bool m0 = gen_prune(nd.x0min, nd.x0max, nd.y0min, nd.y0max, nd.z0min, nd.z0max, gr, closest, te0);
                    const bool m1 = gen_prune(nd.x1min, nd.x1max, nd.y1min, nd.y1max, nd.z1min, nd.z1max, gr, closest, te1);
                    const uint32_t c0 = nd.c0, c1 = nd.c1;
                    const bool both = !m0 && !m1, none = m0 && m1;
                    const bool first0 = te0 <= te1;
                    uint32_t next = both ? (first0 ? c0 : c1) : (m0 ? c1 : c0);
                    if (both) tstack[sp * BLOCK] = (unsigned short)(first0 ? c1 : c0);
                    sp += both ? 1 : 0;
                    const bool more = !none || sp > 0;
                    if (none && sp > 0) { sp--; next = tstack[sp * BLOCK]; }
                    if (!more) state = W_F;
                    else { node = next & 0x7fffu; if (next & 0x8000u) state = W_L; }
                }
                keep = __popcll(__ballot(state == W_T));
            } while (keep >= t_keep);
            WGPROFC(0);
        }
        if (pick == W_L) { /* sequential ifs, not a chain: no register copies at a common join (mega_bvh.h) */
            WGPROF(1, nL);
            uint32_t lpos = 0;
            int cnt = 0;
            if (state == W_L) { lpos = GLEAF_FIRST(node); cnt = (int)GLEAF_COUNT(node); } /* a leaf reference is a run of entries */
            uint32_t e0 = 0, e1 = 0; /* two primitives per trip through two register sets, the next record requested before this one is tested (mega_gen.hip) */
            PrimRec r0, r1;
            { const v4f_t z = {0.f, 0.f, 0.f, 0.f}; r0.a = r0.b = r0.c = r0.d = r0.e = z; r1 = r0; }
            if (cnt > 0) { e0 = entries[lpos]; r0 = load_prim<PRIMS_LDS>(spheres, quads, e0); }
            while (__ballot(cnt > 0) != 0ull) {
                if (cnt > 0) {
                    if (cnt > 1) { e1 = entries[lpos + 1]; r1 = load_prim<PRIMS_LDS>(spheres, quads, e1); }
                    /* both own roots against the same closest_so_far (independent chains), then applied in entry order: dev_gen.h gen_prim_t */
                    const uint32_t ea = e0, eb = e1;
                    const float ta = gen_prim_t(lsc, chains, rec_sphere(r0), rec_quad(r0), ea, ray, ray_a, closest);
                    float tb = -1.0f;
                    if (cnt > 1) tb = gen_prim_t(lsc, chains, rec_sphere(r1), rec_quad(r1), eb, ray, ray_a, closest);
                    if (cnt > 2) { e0 = entries[lpos + 2]; r0 = load_prim<PRIMS_LDS>(spheres, quads, e0); }
                    gen_apply_t(ranks, n_spheres, ta, ea, closest, best, flags);
                    gen_apply_t(ranks, n_spheres, tb, eb, closest, best, flags);
                    lpos += 2; cnt -= 2;
                }
            }
            if (state == W_L) {
                if (sp > 0) {
                    sp--;
                    const uint32_t next = tstack[sp * BLOCK];
                    node = next & 0x7fffu;
                    state = (next & 0x8000u) ? W_L : W_T;
                } else state = W_F;
            }
            WGPROFC(1);
        }
        if (pick == W_F) {
            WGPROF(2, nF);
            /* wave-uniform control flow: every lane runs this block */
            const bool inF = (state == W_F);
            const bool fin = inF && have;
            if (fin && flags) { /* non-ordinary reciprocal / NaN root: the scan decides */
                const WScanHit h = wf_scan_solids(&wp->g.f.r.sc, wp->g.first_medium, (const int *)(wp->g.f.hot_src + wp->g.o_chains), wp->g.n_chains,
                                                  ray.o.x, ray.o.y, ray.o.z, ray.d.x, ray.d.y, ray.d.z, ray.tm);
                best = h.best; closest = h.closest;
            }
            /* ---- retire: hit record at the ray's position, position into its class queue ---- */
            int cls = -1;
            if (fin) {
                WfHit h; h.t = closest; h.best = (int)best;
                hits[pos] = h;
                if (best == GBEST_NONE) cls = WC_FIN;
                else {
                    const uint32_t mat = GENT_QUAD(best) ? quads[GENT_IDX(best)].mat : spheres[GENT_IDX(best)].mat;
                    const int mt = DREF_TYPE(mat);
                    cls = (mt == MORT_MAT_LAMBERTIAN || mt == MORT_MAT_ISOTROPIC) ? WC_LAMB : WC_SPEC;
                }
            }
            const unsigned f_pos = pos;
            if (inF) have = false;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const unsigned long long m = __ballot(cls == k);
                const int n = __popcll(m);
                if (n > 0) {
                    if (staged[k] + n > WG_STAGE) WG_FLUSH(k);
                    const int rk = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                    if (cls == k) stage[k * WG_STAGE + staged[k] + rk] = f_pos;
                    staged[k] += n;
                }
            }
            /* ---- hand the waiting lanes the next records of this wave's batches ---- */
            const unsigned long long need = __ballot(inF);
            const int want = __popcll(need);
            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(need >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need, 0));
            int served = 0;
            while (served < want) {
                if (b_off == b_cnt) { /* batch used up: the next one of this wave */
                    b_base = next_batch * 64u;
                    b_cnt = (b_base < n_items) ? (int)((n_items - b_base < 64u) ? (n_items - b_base) : 64u) : 0;
                    b_off = 0;
                    next_batch += n_waves;
                    if (b_cnt == 0) break;
                }
                int take = b_cnt - b_off;
                if (take > want - served) take = want - served;
                if (inF && rank >= served && rank < served + take) {
                    pos = b_base + (unsigned)(b_off + (rank - served));
                    const WfRay rr = front[pos];
                    ray.o = mk(rr.ox, rr.oy, rr.oz); ray.d = mk(rr.dx, rr.dy, rr.dz); ray.tm = rr.tm;
                    ray_a = vlen2(ray.d);
                    const bool ordinary = gen_ray_setup(ray, g_x, g_y, g_z, g_R, g_mnear, g_kmin, gr);
                    closest = __builtin_inff(); best = GBEST_NONE; sp = 0;
                    flags = ordinary ? 0 : GFL_REF;
                    have = true;
                    if (!ordinary || root == 0xffffu) state = W_F; /* retired at the next F step (the scan decides) */
                    else { node = root & 0x7fffu; state = (root & 0x8000u) ? W_L : W_T; }
                }
                b_off += take; served += take;
            }
            if (inF && !have) state = W_DONE; /* this wave's batches are exhausted */
            WGPROFC(2);
        }
    }
    WG_FLUSH(0); WG_FLUSH(1); WG_FLUSH(2);
#ifdef MORT_PROFILE_STATES
    if (lane == 0) {
        unsigned long long *const counters = wp->g.f.r.counters;
        for (int k = 0; k < 3; k++) { atomicAdd(&counters[4 + k], pr_steps[k]); atomicAdd(&counters[8 + k], pr_lanes[k]); }
        for (int k = 0; k < 4; k++) atomicAdd(&counters[12 + k], pr_cyc[k]);
        atomicAdd(&counters[20], __builtin_amdgcn_s_memrealtime() - rt0); /* sum of wave lifetimes, 10 ns ticks */
        atomicAdd(&counters[21], 1ull);
    }
#endif
#undef WG_FLUSH
}

/* ---- shading of one front ---- */
__global__ void __launch_bounds__(256, MORT_WF_SHADE_WAVES) wf_shade_gen(const WfGenArgs w) {
    const GenArgs &ga = w.g;
    const RenderArgs &a = ga.f.r;
    const DScene &sc = a.sc;
    const int *chains = (const int *)(ga.f.hot_src + ga.o_chains);
    const int par = w.parity;
    const unsigned n0 = w.cnt->cls_count[par][0], n1 = w.cnt->cls_count[par][1], n2 = w.cnt->cls_count[par][2];
    const unsigned b0 = (n0 + 255u) >> 8, b1 = (n1 + 255u) >> 8, b2 = (n2 + 255u) >> 8;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        /* the other parity's class counters were last read by the previous wf_shade_gen; this front's wf_trav_gen is done */
        w.cnt->cls_count[par ^ 1][0] = 0; w.cnt->cls_count[par ^ 1][1] = 0; w.cnt->cls_count[par ^ 1][2] = 0;
    }
    unsigned b = blockIdx.x;
    int cls;
    unsigned n;
    if (b < b0) { cls = WC_LAMB; n = n0; }
    else if (b < b0 + b1) { cls = WC_SPEC; n = n1; b -= b0; }
    else if (b < b0 + b1 + b2) { cls = WC_FIN; n = n2; b -= b0 + b1; }
    else return;
    const unsigned i = b * 256u + threadIdx.x;
    const bool valid = i < n;
    unsigned id = 0;
    bool cont = false; /* path continues into the next front */
    Ray ray; ray.o = mk(0, 0, 0); ray.d = mk(0, 0, 1); ray.tm = 0;
    float out_time0 = 0;
    if (valid) {
        const unsigned pos = w.q_cls[cls][i];
        id = w.q_id[par][pos];
        const WfRay rr = w.q_ray[par][pos];
        const WfHit h = w.hits[pos];
        WfPix P = w.pix[id];
        Rng rng = wf_rng_load(P);
        int s_i = (int)(P.packed & 0xfffu), s_j = (int)((P.packed >> 12) & 0xfffu), iter = (int)(P.packed >> 24);
        ray.o = mk(rr.ox, rr.oy, rr.oz); ray.d = mk(rr.dx, rr.dy, rr.dz); ray.tm = rr.tm;
        const float time0 = rr.time0;
        out_time0 = time0;
        /* the constant media come after every solid, with the solids' closest_so_far (world.cuh:154-160) */
        float closest = h.t;
        uint32_t best = (uint32_t)h.best;
        gen_media(sc, ga.first_medium, sc.n_items, ray, rng, closest, best);
        V3 final_value = a.background; /* a miss (camera.cuh:154-158) */
        bool terminated = (best == GBEST_NONE);
        if (!terminated) {
            const Best bb = gen_decode_best(sc, chains, best, closest);
            /* inlined: this kernel waits on memory (71 % of its wave cycles), and the out-of-line call's frame -- callee-saved registers and
             * the 22-dword result through scratch -- was half of its memory instructions: 2048x2048x4 spp 223 -> 112 ms.  (The call was a
             * workaround for the SLP vectorizer's miscompile, dev_shade_call.h; the library is built without that pass now.  -DMORT_SHADE_CALL
             * brings the call back; mega_gen.hip keeps it, where it costs nothing.) */
#ifdef MORT_SHADE_CALL
            const ShadeOut so = shade_hit_outlined(&a.sc, a.light_type, a.light_idx, ray, time0, bb, rng);
#else
            const ShadeOut so = shade_hit(a.sc, a.light_type, a.light_idx, ray, time0, bb, rng);
#endif
            if (so.done) { final_value = so.final_value; terminated = true; }
            else {
                float4 e; e.x = so.e.kx; e.y = so.e.ky; e.z = so.e.kz; e.w = so.e.rp; /* dielectric: (1, 1, 1, 1) */
                if (so.ident) e.x = e.y = e.z = e.w = 1.0f;
                w.stack[(size_t)iter * (size_t)w.n_paths + id] = e;
                iter++;
                if (iter >= a.bounce_limit) { final_value = mk(0, 0, 0); terminated = true; } /* camera.cuh:161-163 */
                else { P.segments++; cont = true; }
            }
        }
        if (terminated) { /* unwind + accumulate (camera.cuh:165-173,190), then the next sample or the finished pixel */
            while (iter > 0) {
                iter--;
                const float4 e = w.stack[(size_t)iter * (size_t)w.n_paths + id];
                const V3 t = vmul(mk(e.x, e.y, e.z), final_value);
                final_value = vadd(mk(0, 0, 0), vscale(e.w, t));
            }
            P.cr += final_value.x; P.cg += final_value.y; P.cb += final_value.z;
            s_i++;
            if (s_i >= a.sqrt_spp) { s_i = 0; s_j++; }
            const int ly = (int)id / a.width, x = (int)id - ly * a.width;
            if (s_j < a.sqrt_spp) {
                const int y = global_row(ly, a.rank, a.nranks, a.rows_per_block);
                ray = get_ray(a, x, y, rng, s_i, s_j);
                out_time0 = ray.tm;
                P.segments++;
                cont = true;
            } else { /* camera.cuh:194-207 */
                V3 col = vscale(a.pixel_samples_scale, mk(P.cr, P.cg, P.cb));
                if (col.x != col.x) col.x = 0.0f;
                if (col.y != col.y) col.y = 0.0f;
                if (col.z != col.z) col.z = 0.0f;
                if (a.accum) { a.accum[3 * id] = col.x; a.accum[3 * id + 1] = col.y; a.accum[3 * id + 2] = col.z; }
                float g[3] = {mort_sqrtf(col.x), mort_sqrtf(col.y), mort_sqrtf(col.z)};
                unsigned char bq[3];
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    float v = g[k];
                    if (v < 0.0f) v = 0.0f;
                    if (v > 0.999f) v = 0.999f;
                    bq[k] = (unsigned char)mort_f2i(256 * v);
                }
                uchar4 out; out.x = bq[0]; out.y = bq[1]; out.z = bq[2]; out.w = 255;
                a.rgba[id] = out;
                if (a.seg_px) a.seg_px[id] = P.segments;
                mort_rng_state st;
                st.d = rng.d; st.v[0] = rng.v0; st.v[1] = rng.v1; st.v[2] = rng.v2; st.v[3] = rng.v3; st.v[4] = rng.v4;
                st.boxmuller_flag = 0; st.boxmuller_flag_double = 0; st.boxmuller_extra = 0.f; st.boxmuller_extra_double = 0.;
                a.states[id] = st;
                atomicAdd(&a.counters[0], (unsigned long long)P.segments);
                atomicAdd(&a.counters[1], (unsigned long long)rng.draws);
                cont = false;
            }
            if (cont) iter = 0;
        }
        if (cont) {
            wf_rng_store(P, rng);
            P.packed = (uint32_t)s_i | ((uint32_t)s_j << 12) | ((uint32_t)iter << 24);
            w.pix[id] = P;
        }
    }
    /* append the survivors to the next front: one global atomicAdd per workgroup, coalesced record writes */
    {
        __shared__ unsigned s_cnt[4], s_fin[4], s_base;
        const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const unsigned long long m = __ballot(cont);
        const unsigned long long mf = __ballot(valid && !cont);
        if (lane == 0) { s_cnt[wv] = (unsigned)__popcll(m); s_fin[wv] = (unsigned)__popcll(mf); }
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
            const unsigned fin = s_fin[0] + s_fin[1] + s_fin[2] + s_fin[3];
            s_base = tot ? atomicAdd(&w.cnt->front_count[par ^ 1], tot) : 0u;
            if (fin) atomicSub(&w.cnt->live, fin); /* only finished pixels change the live count */
        }
        __syncthreads();
        unsigned off = s_base;
        for (int k = 0; k < wv; k++) off += s_cnt[k];
        const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
        if (cont) {
            const unsigned np = off + (unsigned)rank;
            WfRay o; o.ox = ray.o.x; o.oy = ray.o.y; o.oz = ray.o.z; o.tm = ray.tm; o.dx = ray.d.x; o.dy = ray.d.y; o.dz = ray.d.z; o.time0 = out_time0;
            w.q_ray[par ^ 1][np] = o;
            w.q_id[par ^ 1][np] = id;
        }
    }
}

/* ---- host side ---- */
typedef void (*trav_kernel_t)(const WfGenArgs);
static trav_kernel_t pick_trav(int block, bool prims_in_lds) {
    if (block == 1024) return prims_in_lds ? wf_trav_gen<1024, true> : wf_trav_gen<1024, false>;
    if (block == 512) return prims_in_lds ? wf_trav_gen<512, true> : wf_trav_gen<512, false>;
    return prims_in_lds ? wf_trav_gen<256, true> : wf_trav_gen<256, false>;
}
static int trav_block_for(const GenArgs &ga) { /* the LDS image + per-wave staging must fit one CU's LDS */
    const size_t fixed = (size_t)ga.f.hot_bytes + 32;
    const size_t need512 = fixed + (size_t)MORT_OWN_STACK * 512 * 2 + (size_t)(512 / 64) * 3 * WG_STAGE * 4;
    const size_t need1024 = fixed + (size_t)MORT_OWN_STACK * 1024 * 2 + (size_t)(1024 / 64) * 3 * WG_STAGE * 4;
    int tb = need512 <= 150 * 1024 ? 512 : 256;
    /* a big image (the final scene's 98 KB) admits one workgroup per CU: the kernel needs under 128 VGPRs, so that workgroup can bring four waves per SIMD */
    if (need512 > 75 * 1024 && need1024 <= 158 * 1024) tb = 1024;
    if (const char *e = std::getenv("MORT_WAVE_TRAV_BLOCK")) { const int v = std::atoi(e); if (v == 256 || v == 512 || (v == 1024 && need1024 <= 158 * 1024)) tb = v; }
    return tb;
}
const void *mort_wave_gen_trav_kernel(bool prims_in_lds, int *block) {
    if (block) *block = 512;
    return (const void *)pick_trav(512, prims_in_lds);
}

#define WCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return e_; } while (0)

hipError_t mort_wave_gen_render(const GenArgs &ga, const WfGenHost &hb, int bounce_limit, int sqrt_spp, hipStream_t s, unsigned *live_left) {
    const RenderArgs &a = ga.f.r;
    const size_t N = (size_t)a.width * (size_t)a.local_rows;
    if (live_left) *live_left = 0;
    if (N == 0) return hipSuccess;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
    const size_t o_ray0 = take(N * sizeof(WfRay)), o_ray1 = take(N * sizeof(WfRay));
    const size_t o_id0 = take(N * sizeof(unsigned)), o_id1 = take(N * sizeof(unsigned));
    const size_t o_hits = take(N * sizeof(WfHit)), o_pix = take(N * sizeof(WfPix));
    const size_t o_stack = take(N * (size_t)(bounce_limit > 0 ? bounce_limit : 1) * sizeof(float4));
    const size_t o_c0 = take(N * sizeof(unsigned)), o_c1 = take(N * sizeof(unsigned)), o_c2 = take(N * sizeof(unsigned));
    const size_t o_cnt = take(sizeof(WfCounters));
    const size_t total = off;
    if (*hb.wf_bytes < total) {
        if (*hb.d_wf) { hipFree(*hb.d_wf); *hb.d_wf = nullptr; *hb.wf_bytes = 0; }
        WCHK(hipMalloc(hb.d_wf, total));
        *hb.wf_bytes = total;
    }
    if (!*hb.h_live) WCHK(hipHostMalloc((void **)hb.h_live, 64));
    unsigned char *base = (unsigned char *)*hb.d_wf;
    WfGenArgs w;
    std::memset(&w, 0, sizeof w);
    w.g = ga;
    w.n_paths = (int)N;
    w.q_ray[0] = (WfRay *)(base + o_ray0); w.q_ray[1] = (WfRay *)(base + o_ray1);
    w.q_id[0] = (unsigned *)(base + o_id0); w.q_id[1] = (unsigned *)(base + o_id1);
    w.hits = (WfHit *)(base + o_hits); w.pix = (WfPix *)(base + o_pix);
    w.stack = (float4 *)(base + o_stack);
    w.q_cls[0] = (unsigned *)(base + o_c0); w.q_cls[1] = (unsigned *)(base + o_c1); w.q_cls[2] = (unsigned *)(base + o_c2);
    w.cnt = (WfCounters *)(base + o_cnt);

    const int TB = trav_block_for(ga);
    trav_kernel_t trav = pick_trav(TB, ga.prims_in_lds != 0);
    /* LDS of wf_trav_gen: image | traversal stacks | class staging */
    w.g.f.off_tstack = (ga.f.hot_bytes + 15u) & ~15u;
    w.t_stage = w.g.f.off_tstack + (uint32_t)MORT_OWN_STACK * (uint32_t)TB * 2u;
    const size_t trav_lds = (size_t)w.t_stage + (size_t)(TB / 64) * 3 * WG_STAGE * sizeof(unsigned);
    { /* dynamic + the kernel's own __shared__ objects must fit the CU's 160 KB: refuse here rather than fail at the first launch */
        hipFuncAttributes fattr;
        WCHK(hipFuncGetAttributes(&fattr, (const void *)trav));
        if (trav_lds + fattr.sharedSizeBytes > (size_t)160 * 1024) return hipErrorInvalidValue;
    }
    WCHK(hipFuncSetAttribute((const void *)trav, hipFuncAttributeMaxDynamicSharedMemorySize, (int)trav_lds));
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, trav, TB, trav_lds) != hipSuccess || per_cu < 1) per_cu = 1;
    const int max_trav_grid = hb.num_cus * per_cu;

    const int nb256 = (int)((N + 255) / 256);
    hipLaunchKernelGGL(wf_init_gen, dim3(nb256), dim3(256), 0, s, w);
    WCHK(hipGetLastError());
    const long long max_fronts = (long long)sqrt_spp * sqrt_spp * ((long long)bounce_limit + 1) + 8;
    size_t live = N;
    long long front = 0;
    const int chunk = 32; /* fronts per host round trip (the live count is read back in between) */
    while (live > 0 && front < max_fronts) {
        /* a wave's share of a front: one 64-record batch at least (two were 4 % slower on small fronts: the front's time is its slowest wave's) */
        int tg = (int)((live + (size_t)(TB / 64) * 64 - 1) / ((size_t)(TB / 64) * 64));
        if (tg > max_trav_grid) tg = max_trav_grid;
        if (tg < 1) tg = 1;
        const int sg = (int)((live + 255) / 256) + 3;
        for (int k = 0; k < chunk; k++, front++) {
            w.parity = (int)(front & 1);
            hipLaunchKernelGGL(trav, dim3(tg), dim3(TB), trav_lds, s, w);
            hipLaunchKernelGGL(wf_shade_gen, dim3(sg), dim3(256), 0, s, w);
        }
        WCHK(hipGetLastError());
        WCHK(hipMemcpyAsync(*hb.h_live, &w.cnt->live, sizeof(unsigned), hipMemcpyDeviceToHost, s));
        WCHK(hipStreamSynchronize(s));
        live = **hb.h_live;
    }
    *hb.fronts = (int)front;
    if (live_left) *live_left = (unsigned)live;
    return live == 0 ? hipSuccess : hipErrorUnknown;
}
