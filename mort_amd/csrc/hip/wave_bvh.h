/*
 * wave_bvh.h -- MORT_MODE_WAVE: wavefront (queued) form of the render path for
 * BVH scenes (one BVH of spheres, no light object).
 *
 * One path per pixel, all pixels of the rank in flight.  Each pixel's samples
 * stay chained through its own XORWOW state (the reference's streams are
 * pixel-serial, rng.cuh:17-23), so a "front" is one path segment of every
 * live pixel.  A front is a COMPACT array of (path id, ray) records in HBM;
 * per front, two launches:
 *
 *   wf_trav   world::hit for every record of the front.  Scene hot blob in
 *             LDS.  Each wave owns a contiguous slice of the front and streams
 *             it through two 64-record register buffers (coalesced 32-byte
 *             loads, issued one batch ahead); a lane whose ray is done takes
 *             the next record by cross-lane shuffle (rank among the waiting
 *             lanes -> buffer slot), so the box-test loop runs with nearly all
 *             lanes busy and no load sits on the refill path.  Retiring a ray
 *             writes its 8-byte hit record at the ray's position and appends
 *             that position to the queue of its shade class, staged per wave
 *             in LDS and flushed with one global atomicAdd per ~190 entries.
 *   wf_shade  one workgroup = 256 positions of ONE class: diffuse
 *             (lambertian / isotropic), specular (metal / dielectric /
 *             emissive) or finish (miss: unwind the bounce stack, accumulate,
 *             next sample's camera ray, or write the finished pixel).  No
 *             material divergence inside a wave.  Surviving paths are appended
 *             to the next front (one atomicAdd per workgroup, coalesced
 *             32-byte ray writes).
 *
 * HBM records: front ray 32 B + id 4 B (two parities), hit 8 B (by position),
 * pixel 48 B (by id: XORWOW words, colour sum, packed counters), bounce stack
 * [depth][id] 16 B.  Algorithmic traffic per segment: trav 32 r + 8 w + 4 q;
 * shade 4 q + 4 + 32 + 8 + 48 r, 4 + 32 + 48 + 16 w = 240 B.
 * Counters are double-buffered by front parity and zeroed by the kernel that
 * runs between their last reader and next writer, so a front needs no memset.
 */
#ifndef MORT_WAVE_BVH_H
#define MORT_WAVE_BVH_H

#include "mega_bvh.h" /* own-tree walk: own_prune, slab_check, reference_walk */
#include "wave_common.h"

struct WfArgs {
    RenderArgs r;
    uint32_t off_ring;        /* LDS offset of the per-wave prefetch rings (wf_trav) */
    /* wf_trav's own LDS image (binary own tree, reference leaf nodes, spheres) */
    const unsigned char *trav_src; uint32_t trav_bytes;
    uint32_t t_nodes2, t_leaves, t_spheres, t_tstack, t_stage;
    int node_first, node_count;
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

/* ---- front 0: load streams, first camera ray of every pixel ---- */
extern "C" __global__ void __launch_bounds__(256) wf_init(const WfArgs w) {
    const RenderArgs &a = w.r;
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

/* ---- traversal of one front ---- */
enum { W_T = 0, W_L = 1, W_F = 2, W_DONE = 3 };

#ifndef MORT_WF_TH_F
#define MORT_WF_TH_F 32
#endif
#ifndef MORT_WF_TH_L
#define MORT_WF_TH_L 24
#endif
#ifndef MORT_WF_T_KEEP
#define MORT_WF_T_KEEP 16
#endif
#ifndef MORT_WF_BLOCK
#define MORT_WF_BLOCK 512
#endif
#ifndef MORT_WF_STAGE
#define MORT_WF_STAGE 128 /* positions per class staged in LDS per wave before a flush */
#endif

template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) wf_trav(const WfArgs w) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
#ifdef MORT_PROFILE_STATES
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    {
        const uint4 *src = (const uint4 *)w.trav_src;
        uint4 *dst = (uint4 *)lds;
        const uint32_t n16 = w.trav_bytes >> 4;
        for (uint32_t i = threadIdx.x; i < n16; i += BLOCK) dst[i] = src[i];
    }
    const int par = w.parity;
    if (blockIdx.x == 0 && threadIdx.x == 0) w.cnt->front_count[par ^ 1] = 0; /* next front: its last reader was the previous wf_trav */
    __syncthreads();
    const DNode2 *nodes2 = (const DNode2 *)(lds + w.t_nodes2);
    const DBvhNode *leaves = (const DBvhNode *)(lds + w.t_leaves);
    const DSphere *spheres = (const DSphere *)(lds + w.t_spheres);
    unsigned short *tstack = (unsigned short *)(lds + w.t_tstack) + threadIdx.x; /* [level * BLOCK] */
    unsigned *stage = (unsigned *)(lds + w.t_stage) + (threadIdx.x >> 6) * (3 * MORT_WF_STAGE);
    int staged[3] = {0, 0, 0};
    const int lane = threadIdx.x & 63;
    const int node_first = w.node_first, node_end = w.node_first + w.node_count;
    const unsigned n_items = w.cnt->front_count[par];
    const WfRay *front = w.q_ray[par];

    /* this wave's share of the front: 64-record batches wave_id, wave_id + n_waves, ... (strided, so every
     * wave samples the whole image: fronts are in pixel order and cost varies by region) */
    const unsigned n_waves = gridDim.x * (BLOCK / 64), wave_id = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    unsigned next_batch = wave_id;
    /* Prefetch ring: two batches of 64 records per wave in LDS, filled by LDS-DMA (global_load_lds_dwordx4: no
     * VGPR destination, so nothing forces an early wait).  plane 0 = {ox,oy,oz,tm}, plane 1 = {dx,dy,dz,time0}. */
    float4 *ring = (float4 *)(lds + w.off_ring) + (threadIdx.x >> 6) * 256; /* [buf][plane][64] */
    unsigned A_base = 0, B_base = 0;
    int A_cnt = 0, A_head = 0, B_cnt = 0, cur = 0;
#define WF_DMA(BUFIDX, BASE, CNT) do { \
        BASE = next_batch * 64u; \
        CNT = (BASE < n_items) ? (int)((n_items - BASE < 64u) ? (n_items - BASE) : 64u) : 0; \
        next_batch += n_waves; \
        if (CNT > 0) { \
            const unsigned i_ = BASE + (unsigned)((lane < CNT) ? lane : 0); \
            const float4 *g_ = (const float4 *)(front + i_); \
            __builtin_amdgcn_global_load_lds((const void *)g_, (__attribute__((address_space(3))) void *)(ring + (BUFIDX) * 128), 16, 0, 0); \
            __builtin_amdgcn_global_load_lds((const void *)(g_ + 1), (__attribute__((address_space(3))) void *)(ring + (BUFIDX) * 128 + 64), 16, 0, 0); \
        } } while (0)
    WF_DMA(0, A_base, A_cnt);
    WF_DMA(1, B_base, B_cnt);
    bool ring_fresh = true; /* the first read of a batch needs its DMA to have landed */

    int state = W_F;
    bool have = false;
    unsigned pos = 0;
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 1, tm = 0;
    OwnRay orr; orr.ix = orr.iy = orr.iz = 1; orr.mx = orr.my = orr.mz = 0; orr.band = 0; orr.invlen = 1;
    float ra = 1, closest = 0;
    int best = -1;          /* sphere | leaf << 16 */
    uint32_t node = 0;      /* T: own-tree node; L: leaf record */
    int sp = 0, flags = 0;
    uint32_t bmat = 0;

#define WF_FLUSH(k) do { \
        if (staged[k] > 0) { \
            unsigned base_ = 0; \
            if (lane == 0) base_ = atomicAdd(&w.cnt->cls_count[par][k], (unsigned)staged[k]); \
            base_ = __shfl(base_, 0); \
            for (int i_ = lane; i_ < staged[k]; i_ += 64) w.q_cls[k][base_ + (unsigned)i_] = stage[(k) * MORT_WF_STAGE + i_]; \
            staged[k] = 0; \
        } } while (0)

#ifdef MORT_PROFILE_STATES
    unsigned long long pr_steps[3] = {0, 0, 0}, pr_lanes[3] = {0, 0, 0}, pr_cyc[4] = {0, 0, 0, 0};
    unsigned long long pt0 = __builtin_readcyclecounter(), pt1;
#define WPROF(i, lanes) do { pr_steps[i] += 1; pr_lanes[i] += (unsigned long long)(lanes); } while (0)
#define WPROFC(i) do { pt1 = __builtin_readcyclecounter(); pr_cyc[i] += pt1 - pt0; pt0 = pt1; } while (0)
#else
#define WPROF(i, lanes) do { } while (0)
#define WPROFC(i) do { } while (0)
#endif

    for (;;) {
        const int nT = __popcll(__ballot(state == W_T));
        const int nL = __popcll(__ballot(state == W_L));
        const int nF = __popcll(__ballot(state == W_F));
        if (nT + nL + nF == 0) break;
        int pick;
        if (nF >= MORT_WF_TH_F) pick = W_F;
        else if (nL >= MORT_WF_TH_L) pick = W_L;
        else if (nT > 0) pick = W_T;
        else pick = (nL >= nF) ? W_L : W_F;
        WPROFC(3);

        if (pick == W_T) {
            int keep;
            do {
                WPROF(0, __popcll(__ballot(state == W_T)));
                if (state == W_T) { /* both child boxes of one node of the own tree (mega_bvh.h) */
                    const float4 *np = (const float4 *)(nodes2 + node);
                    const float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
                    float te0, te1;
                    const bool m0 = own_prune(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q3.z, orr, closest, te0);
                    const bool m1 = own_prune(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.w, orr, closest, te1);
                    const uint32_t c0 = __float_as_uint(q3.x), c1 = __float_as_uint(q3.y);
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
            } while (keep >= MORT_WF_T_KEEP);
            WPROFC(0);
        }
        if (pick == W_L) { /* sequential ifs, not a chain: no register copies at a common join (mega_bvh.h) */
            WPROF(1, nL);
            if (state == W_L) { /* sphere::hit on the spheres of a reference leaf node (objects.cuh:60-77,690-692) */
                const uint32_t leaf = leaves[node].prims;
                const uint32_t pa = leaf & 0x7fffu, pb = (leaf >> 16) & 0x7fffu;
                Ray r; r.o = mk(ox, oy, oz); r.d = mk(dx, dy, dz); r.tm = tm;
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const uint32_t p = k ? pb : pa;
                    if (k == 1 && pb == pa) break;
                    const float t = sphere_hit_root(spheres[p], r, ra, 0.001f, closest);
                    if (t != -1.0f) {
                        if (t == closest && best >= 0) flags |= FL_TIE;
                        closest = t; best = (int)(p | (node << 16));
                    }
                }
                if (sp > 0) {
                    sp--;
                    const uint32_t next = tstack[sp * BLOCK];
                    node = next & 0x7fffu;
                    state = (next & 0x8000u) ? W_L : W_T;
                } else state = W_F;
            }
            WPROFC(1);
        }
        if (pick == W_F) {
            WPROF(2, nF);
            /* wave-uniform control flow: every lane runs this block.  The refill comes first so that the
             * wait for the prefetched batch never covers the stores issued by this step's retire. */
            const bool inF = (state == W_F);
            const bool fin = inF && have;
            if (fin) { /* is the winner what bvh_node::hit returns?   */
                Ray r; r.o = mk(ox, oy, oz); r.d = mk(dx, dy, dz); r.tm = tm;
                bool need_ref = flags != 0;
                if (!need_ref && best >= 0) need_ref = !slab_check(leaves[best >> 16], r, orr, closest);
                if (need_ref) {
                    const RefHit h = reference_walk(w.r.sc.nodes, node_first, node_end, spheres, ox, oy, oz, dx, dy, dz, tm, ra);
                    best = h.best; closest = h.closest;
                }
                if (best >= 0) { best &= 0x7fff; bmat = spheres[best].mat; }
            }
            const unsigned f_pos = pos;
            const float f_t = closest;
            const int f_best = best;
            const uint32_t f_bmat = bmat;
            if (inF) have = false;
            /* ---- hand the waiting lanes the next records of the slice ---- */
            const unsigned long long need = __ballot(inF);
            const int want = __popcll(need);
            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(need >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need, 0));
            int served = 0;
#pragma unroll
            for (int round = 0; round < 2; round++) {
                if (served < want) {
                    if (A_head == A_cnt && B_cnt > 0) { /* batch used up: switch to the prefetched one, start loading the one after */
                        cur ^= 1; A_base = B_base; A_cnt = B_cnt; A_head = 0;
                        WF_DMA(cur ^ 1, B_base, B_cnt);
                        ring_fresh = true;
                    }
                    int take = A_cnt - A_head;
                    if (take > want - served) take = want - served;
                    if (take > 0) {
                        if (ring_fresh) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); ring_fresh = false; }
                        if (inF && rank >= served && rank < served + take) {
                            const int slot = A_head + (rank - served);
                            const float4 r0 = ring[cur * 128 + slot], r1 = ring[cur * 128 + 64 + slot];
                            ox = r0.x; oy = r0.y; oz = r0.z; tm = r0.w; dx = r1.x; dy = r1.y; dz = r1.z;
                            pos = A_base + (unsigned)slot;
                            ra = dx * dx + dy * dy + dz * dz;
                            orr.ix = 1.0f / dx; orr.iy = 1.0f / dy; orr.iz = 1.0f / dz;
                            orr.mx = ox * orr.ix; orr.my = oy * orr.iy; orr.mz = oz * orr.iz;
                            const float mm = __builtin_fmaxf(__builtin_fmaxf(mort_fabsf(orr.mx), mort_fabsf(orr.my)), mort_fabsf(orr.mz));
                            orr.band = mm * 4.76837158203125e-07f;
                            orr.invlen = 1.01f / mort_sqrtf(ra);
                            const bool ordinary = own_inv_ok(orr.ix) && own_inv_ok(orr.iy) && own_inv_ok(orr.iz) && (mm < 1e30f);
                            closest = __builtin_inff(); best = -1; bmat = 0; node = 0; sp = 0;
                            flags = ordinary ? 0 : FL_REF;
                            have = true;
                            state = ordinary ? W_T : W_F;
                        }
                        A_head += take; served += take;
                    }
                }
            }
            /* ---- retire the finished rays: hit record at the ray's position, position into its class queue ---- */
            int cls = -1;
            if (fin) {
                WfHit h; h.t = f_t; h.best = f_best;
                w.hits[f_pos] = h;
                if (f_best < 0) cls = WC_FIN;
                else { const int mt = DREF_TYPE(f_bmat); cls = (mt == MORT_MAT_LAMBERTIAN || mt == MORT_MAT_ISOTROPIC) ? WC_LAMB : WC_SPEC; }
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const unsigned long long m = __ballot(cls == k);
                const int n = __popcll(m);
                if (n > 0) {
                    if (staged[k] + n > MORT_WF_STAGE) WF_FLUSH(k);
                    const int rk = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                    if (cls == k) stage[k * MORT_WF_STAGE + staged[k] + rk] = f_pos;
                    staged[k] += n;
                }
            }
            if (inF && !have && A_head == A_cnt && B_cnt == 0) state = W_DONE; /* slice exhausted */
            WPROFC(2);
        }
    }
    WF_FLUSH(0); WF_FLUSH(1); WF_FLUSH(2);
#ifdef MORT_PROFILE_STATES
    if (lane == 0) {
        for (int k = 0; k < 3; k++) { atomicAdd(&w.r.counters[4 + k], pr_steps[k]); atomicAdd(&w.r.counters[8 + k], pr_lanes[k]); }
        for (int k = 0; k < 4; k++) atomicAdd(&w.r.counters[12 + k], pr_cyc[k]);
        const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
        atomicAdd(&w.r.counters[20], rt1 - rt0);           /* sum of wave lifetimes, 10 ns ticks */
        atomicAdd(&w.r.counters[21], 1ull);                /* waves */
        atomicAdd(&w.r.counters[22], pt0 - 0ull > 0 ? (unsigned long long)0 : 0ull);
    }
#endif
#undef WF_FLUSH
#undef WF_DMA
}

/* ---- shading of one front ---- */
DEV void wf_texture(const DScene &sc, const DLambert &m, V3 outward, V3 p, V3 &color) {
    color = mk(m.r, m.g, m.b);
    if (m.tex == 0) return;
    float u = 0, v = 0;
    uint32_t tex = m.tex;
    for (int guard = 0; guard < 8; guard++) { /* only image textures / the error pattern read (u, v) */
        const int tt = DREF_TYPE(tex);
        if (tt == MORT_TEXTURE_CHECKER) {
            const DChecker c = sc.checker[DREF_IDX(tex)];
            const int xi = mort_f2i(mort_floorf(c.inv_scale * p.x));
            const int yi = mort_f2i(mort_floorf(c.inv_scale * p.y));
            const int zi = mort_f2i(mort_floorf(c.inv_scale * p.z));
            tex = ((xi + yi + zi) % 2 == 0) ? c.even : c.odd;
            continue;
        }
        if (tt != MORT_TEXTURE_SOLID && tt != MORT_TEXTURE_NOISE) sphere_uv(outward, u, v);
        break;
    }
    color = texture_value(sc, tex, u, v, p);
}

extern "C" __global__ void __launch_bounds__(256, MORT_WF_SHADE_WAVES) wf_shade(const WfArgs w) {
    const RenderArgs &a = w.r;
    const DScene &sc = a.sc;
    const int par = w.parity;
    const unsigned n0 = w.cnt->cls_count[par][0], n1 = w.cnt->cls_count[par][1], n2 = w.cnt->cls_count[par][2];
    const unsigned b0 = (n0 + 255u) >> 8, b1 = (n1 + 255u) >> 8, b2 = (n2 + 255u) >> 8;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        /* the other parity's class counters were last read by the previous wf_shade; this front's wf_trav is done */
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
        V3 final_value = a.background; /* WC_FIN from wf_trav: a miss (camera.cuh:154-158) */
        bool terminated = (cls == WC_FIN);
        if (!terminated) {
            const DSphere sp = sc.spheres[h.best];
            const V3 p = ray_at(ray, h.t);
            const V3 outward = vdiv(vsub(p, sphere_center(sp, ray.tm)), sp.radius);
            const bool front_face = vdot(ray.d, outward) < 0;
            const V3 normal = front_face ? outward : vneg(outward);
            const int mtype = DREF_TYPE(sp.mat), midx = DREF_IDX(sp.mat);
            float4 e;
            e.x = e.y = e.z = e.w = 1.0f;
            V3 ndir = mk(0, 0, 1);
            if (cls == WC_LAMB) { /* materials.cuh:38-55,182-198; pdf.cuh:29-54 */
                const bool lamb = (mtype == MORT_MAT_LAMBERTIAN);
                const DLambert m = lamb ? sc.lambert[midx] : sc.isotropic[midx];
                V3 attenuation;
                wf_texture(sc, m, outward, p, attenuation);
                float mat_pdf, scattering_pdf;
                if (lamb) {
                    const Onb uvw = onb_from_w(normal);
                    ndir = onb_local(uvw, random_cosine_direction(rng));
                    const V3 ud = vunit(ndir);
                    const float cosine_theta = vdot(ud, uvw.w);
                    mat_pdf = mort_fmaxf(0, (float)((double)cosine_theta / 3.1415926));
                    const float cos_theta = vdot(normal, ud);
                    scattering_pdf = (cos_theta < 0) ? 0.0f : (float)((double)cos_theta / 3.141592565);
                } else {
                    ndir = random_unit_vector(rng);
                    mat_pdf = (float)(1 / (4 * 3.1415926));
                    scattering_pdf = (float)(1 / (4 * 3.1415926));
                }
                e.x = scattering_pdf * attenuation.x; e.y = scattering_pdf * attenuation.y; e.z = scattering_pdf * attenuation.z;
                e.w = 1 / mat_pdf;
                ray.tm = time0; /* ray(rec.p, dir, r.time()) (camera.cuh:119) */
            } else if (mtype == MORT_MAT_METAL) { /* materials.cuh:73-84 */
                const DMetal m = sc.metal[midx];
                const V3 reflected = reflect(ray.d, normal);
                ndir = vadd(vunit(reflected), vscale(m.fuzz, random_unit_vector(rng)));
                e.x = 1.0f * m.r; e.y = 1.0f * m.g; e.z = 1.0f * m.b; e.w = 1.0f;
            } else if (mtype == MORT_MAT_DIELECTRIC) { /* materials.cuh:107-130 */
                const DDielectric m = sc.dielectric[midx];
                const float refraction_ratio = front_face ? m.inv_ior : m.ior;
                const V3 unit_direction = vunit(ray.d);
                const float cos_theta = (float)mort_fmin((double)vdot(vneg(unit_direction), normal), 1.0);
                const float sin_theta = (float)mort_sqrt(1.0 - (double)(cos_theta * cos_theta));
                const bool cant_refract = (double)(refraction_ratio * sin_theta) > 1.0;
                if (cant_refract || reflectance(cos_theta, refraction_ratio) > random_float(rng))
                    ndir = reflect(unit_direction, normal);
                else
                    ndir = refract(unit_direction, normal, refraction_ratio);
            } else { /* diffuse_light / unknown tag: no scatter (materials.cuh:151-163) */
                V3 emission = mk(0, 0, 0);
                if (mtype == MORT_MAT_DIFFUSE_LIGHT && front_face) wf_texture(sc, sc.dlight[midx], outward, p, emission);
                final_value = emission;
                terminated = true;
            }
            if (!terminated) {
                w.stack[(size_t)iter * (size_t)w.n_paths + id] = e;
                iter++;
                ray.o = p; ray.d = ndir;
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

#endif
