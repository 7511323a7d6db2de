/*
 * mega_bvh.h -- the hot kernel for BVH scenes (Scene 1, Scene 10): world =
 * one non-skip BVH over spheres, bvh_mode, no light object.
 *
 * CDNA4 design (each point bit-identical to the reference's arithmetic):
 *
 *  - Scene in LDS.  The kernel's image of the scene (its own tree as four-child nodes, a record per reference leaf
 *    node with its one or two spheres by value and the leaf's box, material and texture tables: 55 KB for Scene 1) is
 *    copied into LDS once per workgroup; a box step reads one node with eight independent ds_read_b128, a leaf step
 *    one record with five.  Node and record sizes are an ODD number of 16-byte pieces (144 B, 112 B): the lanes of a
 *    wave read the same piece of different records, and only an odd stride spreads those over all LDS banks.
 *    Frames with two pixels per lane or more run on 1024-thread workgroups compiled for 128 registers: four waves
 *    per SIMD (DESIGN.md 4.1 h).
 *    The reference reads 4 SoA arrays + a 24-byte aabb from global memory per node (objects.cuh:728-731).
 *
 *  - Per-lane state machine, wave-level scheduling.  Each lane owns one pixel
 *    and is in one of three states: T (at a tree node, both child boxes to test),
 *    L (inside a leaf, sphere tests pending), S (closest hit known: shade,
 *    scatter, or finish the sample / pixel and start the next ray).  Each
 *    iteration of the wave's loop runs ONE state's code, chosen from the
 *    ballot/popcount of the lanes in each state, for the lanes in that state.
 *    A lane whose path or pixel ends never waits for the wave's longest path
 *    (the reference's loop nest, camera.cuh:96-159,187-192, runs every lane in
 *    lockstep with the slowest).
 *
 *  - Dynamic pixels.  Lanes fetch pixels from one atomic counter (one atomicAdd per wave per refill, lanes ranked
 *    by mbcnt), most expensive 8x8 tiles first, so the chip stays full until the pool is empty; partitions with
 *    about one pixel per lane spread a fetch over the whole cost order and switch to drain mode (below).
 *
 *  - Own tree, near child first.  The reference walks its median-split tree left-first (objects.cuh:664-723), 41
 *    box tests per segment in Scene 1.  bvh_node::hit returns the closest accepted sphere hit; WHICH one that is
 *    depends on the walk only through the leaf nodes it refuses to enter.  So this kernel walks its own SAH tree
 *    over the reference's leaf nodes (four boxes per step, nearest child next), and proves per ray that the reference's walk
 *    would have returned the same sphere -- or else repeats the ray with the reference's walk:
 *      * leaf records keep the reference leaf box bit for bit, inner boxes are exact unions, so aabb::hit
 *        (aabb.cuh:37-59) passing on a leaf implies it passes on every box above it, in either tree (rounding is
 *        monotonic); the fp32 prune below is conservative against aabb::hit (error band tau);
 *      * a box is skipped "because of closest_so_far" only if it is entered later than closest_so_far by more
 *        than the length of the longest leaf box below it (+ 0.8 %): this cannot hide a sphere hit that lies
 *        BEFORE the entry into its own leaf box (the only kind whose acceptance depends on visiting order);
 *      * the winner (minimum t over all sphere hits seen, ties flagged) is accepted iff aabb::hit of its leaf
 *        box passes with t_max = t, evaluated exactly as the reference does (slab_check).  If not, or on a tie,
 *        or for rays whose reciprocal direction is not an ordinary float, the reference's own walk is run.
 *
 *  - Hit record rebuilt only for the winner; sphere uv (acosf/atan2f) only when
 *    the material's texture reads it.
 */
#ifndef MORT_MEGA_BVH_H
#define MORT_MEGA_BVH_H

#include "dev_render.h"


struct FastArgs {
    RenderArgs r;
    const unsigned char *hot_src; /* device copy of the kernel's LDS image */
    uint32_t hot_bytes;
    uint32_t off_nodes4, off_leafrecs, off_lambert, off_metal, off_diel, off_dlight, off_iso, off_solid, off_checker;
    uint32_t off_tstack;  /* LDS offset of the per-lane traversal stacks: [levels][thread] u16 (four-wide: the world's own bound, at most MORT_OWN4_STACK) */
    int node_first, node_count; /* the reference's threaded nodes in r.sc.nodes (HBM): fallback walk only */
    unsigned int *next_q; /* work counter, zeroed before launch */
    int tiles_x, tiles_total;
    uint32_t off_stack;   /* LDS offset of the per-lane bounce stacks: [depth][thread] float4 */
    int stack_lds_depth;  /* entries per lane kept in LDS; deeper entries spill to private memory */
    const unsigned *tile_order; /* tiles by decreasing cost, or null = row-major */
    int gen_tiles;              /* tiles per generation of the slot -> pixel map (below); 0 = plain order */
    int spread_shift;           /* log2 of the pixels a wave takes from one tile (6 = whole tiles ... 0 = single pixels) */
    unsigned *tile_cost;        /* per tile: segments traced this frame (feeds the next frame's order), or null */
    int th_s, th_l, t_keep; /* scheduling thresholds (lanes): batch sizes that trigger a shade / leaf step, and the
                               lane count below which the box-step loop hands control back (wave-uniform) */
    /* MORT_MODE_THROUGHPUT (NOT the reference's streams): one stream and one work item per (pixel, stratum row).  The launch then
     * covers a VIRTUAL image of local_rows * sub rows: virtual row = row * sub + s_j; r.states / vaccum are indexed by virtual pixel */
    int sub;                    /* 0, or sqrt_spp in sub-stream launches */
    float *vaccum;              /* sub-stream launches: per virtual pixel, the unscaled colour sum of its stratum row (3 floats) */
    float4 *deep;               /* bounce-stack levels >= stack_lds_depth: [level - stack_lds_depth][lane of the launch] (HBM, coalesced per wave) */
    /* priority pixels (mega_gen.hip): the pixels of the first prio_tiles tiles of the cost order are handed out prio_lanes per fetch from
     * their own cursor (next_q[1]), so that every wave holds at most a few of the frame's longest chains, and flagged: their wave runs
     * the state THEY are in next (PixelFetch.prio).  0 = none */
    int prio_tiles, prio_lanes;
    int drain_rounds;           /* DRAIN kernels, once the pool is empty: 1 = run in rounds (every live lane one segment per shade step), 0 = follow the lane furthest behind */
    int tile_key_sum;           /* 1 = order tiles by their segment sum (round 2), 0 = by their longest pixel */
    /* heavy waves (mega_gen.hip; measured there): of every heavy_mod consecutive waves the first heavy_num take pixels with heavy_cap lanes only and take them from
     * the HEAD of the cost order (the first *prio_dev tiles: the frame's longest pixel chains) while it lasts; the other waves start at the tail.  A round of a wave
     * costs less the fewer and the more alike its lanes are (final scene, tail waves: 64 lanes 60 us, 8 lanes 40 us, 4 lanes 31 us per round) */
    const unsigned *prio_dev;   /* device word: tiles in the head (tile_sort.hip heavy_count_kernel), or null */
    int heavy_mod, heavy_num, heavy_cap;
    int lane_cap;               /* lanes of a wave that take pixels (64 = all).  A partition with fewer pixels than the chip has lanes is bound by the latency of its
                                 * waves' steps, not by issue slots: the same pixels on MORE waves with FEWER lanes each shorten every wave's chain of steps */
    unsigned long long *wave_log; /* profile builds (-DMORT_PROFILE_STATES) with MORT_WAVE_LINES=1: 16 words per wave of the launch, else null */
};

#ifdef MORT_BVH_PRIVATE_ARGS
#define MORT_BVH_ARG_NAME fa_byval
#else
#define MORT_BVH_ARG_NAME
#endif
#ifdef MORT_BVH_NO_PIN /* measured variant: uniform values are read where they are used, not pinned in SGPRs */
#define BU_I(x) (x)
#define BU_U(x) (x)
#define BU_F(x) (x)
#define BU_P(x) (x)
#else
#define BU_I(x) uni_i(x)
#define BU_U(x) uni_u(x)
#define BU_F(x) uni_f(x)
#define BU_P(x) uni_p(x)
#endif
/* a wave-uniform value moved into an SGPR: the compiler can neither re-load nor re-materialise it inside the state loop */
__device__ __forceinline__ int uni_i(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ uint32_t uni_u(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ float uni_f(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }
template <typename T> __device__ __forceinline__ T *uni_p(T *p) {
    const unsigned long long v = (unsigned long long)p;
    const uint32_t lo = uni_u((uint32_t)v), hi = uni_u((uint32_t)(v >> 32));
    return (T *)(((unsigned long long)hi << 32) | (unsigned long long)lo);
}

enum { ST_T = 0, ST_L = 1, ST_S = 2, ST_DONE = 3 };
enum { FL_TIE = 1, FL_REF = 2 };
/* -DMORT_REGION_MARKS: comments in the ISA at region boundaries, for static instruction counts per region
 * (scripts/isa_regions.py) */
#ifdef MORT_REGION_MARKS
#define REGION(name) asm volatile("; REGION " name)
#else
#define REGION(name) do { } while (0)
#endif

DEV bool own_inv_ok(float v) { const float a = mort_fabsf(v); return a > 1e-15f && a < 1e15f; } /* !(NaN) too */
/* Textures whose tables stay in HBM (image, noise) and the error pattern, with the sphere uv they read: rare in
 * the scenes this kernel serves, kept out of line so that their registers and code stay out of the shade step */
struct V3Ret { float x, y, z; };
__device__ __attribute__((noinline)) V3Ret texture_value_uv(const DScene *sc, uint32_t tex, float nx, float ny, float nz, float px, float py, float pz) {
    float u, v;
    sphere_uv(mk(nx, ny, nz), u, v);
    const V3 c = texture_value(*sc, tex, u, v, mk(px, py, pz));
    V3Ret r; r.x = c.x; r.y = c.y; r.z = c.z;
    return r;
}

/* bvh_node::hit as the reference walks it (objects.cuh:664-723): its threaded nodes (HBM), left first, aabb::hit
 * evaluated exactly.  Out of line: it runs for about one segment in 10^5. */
struct RefHit { int best; float closest; };
__device__ __attribute__((noinline)) RefHit reference_walk(const DBvhNode *nodes, int node_first, int node_end, const DSphere *spheres,
                                                           float ox, float oy, float oz, float dx, float dy, float dz, float tm, float ray_a) {
    Ray ray; ray.o = mk(ox, oy, oz); ray.d = mk(dx, dy, dz); ray.tm = tm;
    float closest = __builtin_inff();
    int best = -1;
    const double dix = 1.0 / (double)ray.d.x, diy = 1.0 / (double)ray.d.y, diz = 1.0 / (double)ray.d.z;
    int n = node_first;
    while (n < node_end) {
        const DBvhNode nd = nodes[n];
        const int skip = (int)(nd.skip & 0x7fffffffu);
        if (!slab_exact(nd, ray.o.x, ray.o.y, ray.o.z, dix, diy, diz, closest)) { n = skip; continue; }
        if (!(nd.skip >> 31)) { n = n + 1; continue; }
        const uint32_t pa = nd.prims & 0x7fffu, pb = (nd.prims >> 16) & 0x7fffu;
        for (int k = 0; k < 2; k++) {
            const uint32_t p = k ? pb : pa;
            if (k == 1 && pb == pa) break;
            float t;
            if (sphere_hit_t(spheres[p], ray, ray_a, 0.001f, closest, t)) { closest = t; best = (int)p; }
        }
        n = skip;
    }
    RefHit h; h.best = best; h.closest = closest;
    return h;
}

/* sphere::hit (objects.cuh:60-77) returning the accepted root, or -1 for "no hit" (an accepted root is >= t_min > 0 or
 * NaN, never -1): the same operations as sphere_hit_t, without a result variable that is only sometimes written
 * (which the compiler kept in private memory in the middle of the leaf step) */
DEV float sphere_hit_root(const DSphere &s, const Ray &r, float a, float t_min, float t_max) {
    const V3 oc = vsub(r.o, sphere_center(s, r.tm));
    const float half_b = vdot(oc, r.d);
    const float c = vlen2(oc) - s.radius * s.radius;
    const float discriminant = half_b * half_b - a * c;
    if (discriminant < 0) return -1.0f;
    const float sqrtd = mort_sqrtf(discriminant);
    float root = (-half_b - sqrtd) / a;
    if (root < t_min || t_max < root) {
        root = (-half_b + sqrtd) / a;
        if (root < t_min || t_max < root) return -1.0f;
    }
    return root;
}

/* per-ray constants of the own-tree box test: p = b * inv - o * inv */
struct OwnRay { float ix, iy, iz, mx, my, mz, band, invlen; };

/* true = skip the box.  te: entry distance (>= t_min = 0.001).
 * Conservative against aabb::hit at any t_max > closest: |p - reference's (b - o) * inv| <= 2^-21 |p| + 2^-24 |o * inv|,
 * tau = 2^-20 max(|te|, |tx|) + 2^-21 max |o * inv|. */
DEV bool own_prune(float xmin, float xmax, float ymin, float ymax, float zmin, float zmax, float e, const OwnRay &r, float closest, float &te_out) {
    const float px0 = __builtin_fmaf(xmin, r.ix, -r.mx), px1 = __builtin_fmaf(xmax, r.ix, -r.mx);
    const float py0 = __builtin_fmaf(ymin, r.iy, -r.my), py1 = __builtin_fmaf(ymax, r.iy, -r.my);
    const float pz0 = __builtin_fmaf(zmin, r.iz, -r.mz), pz1 = __builtin_fmaf(zmax, r.iz, -r.mz);
    const float te = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(px0, px1), __builtin_fminf(py0, py1)),
                                     __builtin_fmaxf(__builtin_fminf(pz0, pz1), 0.001f));
    const float tx = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(px0, px1), __builtin_fmaxf(py0, py1)), __builtin_fmaxf(pz0, pz1));
    const float tau = __builtin_fmaf(__builtin_fmaxf(mort_fabsf(te), mort_fabsf(tx)), 9.5367431640625e-07f, r.band);
    const float key = __builtin_fmaf(te, 0.992f, -__builtin_fmaf(e, r.invlen, tau));
    te_out = te;
    return (tx - te < -tau) || (key > closest);
}

/* own_prune for two boxes at once, the fused multiply-adds as v_pk_fma_f32 (two per lane and instruction; identical IEEE results).  On MI355X a packed
 * instruction costs a busy SIMD the issue time of two plain ones (calibration kind 4: 4.1 against 2.2 cycles), so a frame that fills the chip gains nothing -- but
 * a wave that has its SIMD almost to itself issues one instruction per ~5 cycles whatever its width, and that is the state of every chain-bound launch */
typedef float v2f_t __attribute__((ext_vector_type(2)));
DEV void own_prune2(v2f_t xmin, v2f_t xmax, v2f_t ymin, v2f_t ymax, v2f_t zmin, v2f_t zmax, v2f_t e, const OwnRay &r, float closest,
                    float &te_a, float &te_b, bool &skip_a, bool &skip_b) {
    const v2f_t ix = {r.ix, r.ix}, iy = {r.iy, r.iy}, iz = {r.iz, r.iz};
    const v2f_t nmx = {-r.mx, -r.mx}, nmy = {-r.my, -r.my}, nmz = {-r.mz, -r.mz};
    const v2f_t px0 = __builtin_elementwise_fma(xmin, ix, nmx), px1 = __builtin_elementwise_fma(xmax, ix, nmx);
    const v2f_t py0 = __builtin_elementwise_fma(ymin, iy, nmy), py1 = __builtin_elementwise_fma(ymax, iy, nmy);
    const v2f_t pz0 = __builtin_elementwise_fma(zmin, iz, nmz), pz1 = __builtin_elementwise_fma(zmax, iz, nmz);
    v2f_t te, tx;
    te.x = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(px0.x, px1.x), __builtin_fminf(py0.x, py1.x)), __builtin_fmaxf(__builtin_fminf(pz0.x, pz1.x), 0.001f));
    te.y = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(px0.y, px1.y), __builtin_fminf(py0.y, py1.y)), __builtin_fmaxf(__builtin_fminf(pz0.y, pz1.y), 0.001f));
    tx.x = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(px0.x, px1.x), __builtin_fmaxf(py0.x, py1.x)), __builtin_fmaxf(pz0.x, pz1.x));
    tx.y = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(px0.y, px1.y), __builtin_fmaxf(py0.y, py1.y)), __builtin_fmaxf(pz0.y, pz1.y));
    v2f_t am;
    am.x = __builtin_fmaxf(mort_fabsf(te.x), mort_fabsf(tx.x)); am.y = __builtin_fmaxf(mort_fabsf(te.y), mort_fabsf(tx.y));
    const v2f_t c20 = {9.5367431640625e-07f, 9.5367431640625e-07f}, band = {r.band, r.band}, il = {r.invlen, r.invlen}, c992 = {0.992f, 0.992f};
    const v2f_t tau = __builtin_elementwise_fma(am, c20, band);
    const v2f_t key = __builtin_elementwise_fma(te, c992, -__builtin_elementwise_fma(e, il, tau));
    const v2f_t gap = tx - te;
    te_a = te.x; te_b = te.y;
    skip_a = (gap.x < -tau.x) || (key.x > closest);
    skip_b = (gap.y < -tau.y) || (key.y > closest);
}

/* aabb::hit (aabb.cuh:37-59) with t_min = 0.001, t_max = closest: the reference's boolean (see slab_hit), the fp64
 * reciprocals computed only for the rare ray inside the fp32 error band.  Caller: the ray has ordinary reciprocals. */
DEV bool slab_check(const DBvhNode &nd, const Ray &ray, const OwnRay &r, float closest) {
    /* r.i* = fl32(1 / d): within one ulp of slab_hit's fl32(1.0 / (double)d), inside its error budget (4u of 16u) */
    const float px0 = (nd.xmin - ray.o.x) * r.ix, px1 = (nd.xmax - ray.o.x) * r.ix;
    const float py0 = (nd.ymin - ray.o.y) * r.iy, py1 = (nd.ymax - ray.o.y) * r.iy;
    const float pz0 = (nd.zmin - ray.o.z) * r.iz, pz1 = (nd.zmax - ray.o.z) * r.iz;
    const float t_min = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(px0, px1), __builtin_fminf(py0, py1)),
                                        __builtin_fmaxf(__builtin_fminf(pz0, pz1), 0.001f));
    const float t_max = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(px0, px1), __builtin_fmaxf(py0, py1)),
                                        __builtin_fminf(__builtin_fmaxf(pz0, pz1), closest));
    const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(mort_fabsf(px0), mort_fabsf(px1)), __builtin_fmaxf(mort_fabsf(py0), mort_fabsf(py1))),
                                    __builtin_fmaxf(mort_fabsf(pz0), mort_fabsf(pz1)));
    const float gap = t_max - t_min;
    const bool decided = (mort_fabsf(gap) > m * 9.5367431640625e-07f) && (m < 1e30f) && (m > 1e-30f);
    bool hit = gap > 0;
    if (!decided) hit = slab_exact(nd, ray.o.x, ray.o.y, ray.o.z, 1.0 / (double)ray.d.x, 1.0 / (double)ray.d.y, 1.0 / (double)ray.d.z, closest);
    return hit;
}
enum { K_SHADE = 0, K_FINISH = 1, K_NEWSAMPLE = 2, K_NEWPIX = 3 };

#ifndef MORT_TH_S
#define MORT_TH_S 48
#endif
#ifndef MORT_TH_L
#define MORT_TH_L 16
#endif
#ifndef MORT_T_UNROLL
#define MORT_T_UNROLL 1 /* box steps per check of the lane count (four-wide steps: 1 measured 111.6 ms, 2 113.0, 3 116.6 on Scene 1) */
#endif
#ifndef MORT_T_KEEP
#define MORT_T_KEEP 16
#endif

#ifndef MORT_MIN_WAVES
#define MORT_MIN_WAVES 3
#endif
#ifndef MORT_FAST_BLOCK
#define MORT_FAST_BLOCK 768
#endif

/* ---- a pixel's end and the next pixel's start: once per 484 samples per lane, kept out of line so that their
 * registers and code stay out of the shade step (like the texture and reference-walk paths above) ---- */
template <bool PROBE, bool SUB = false>
__device__ __attribute__((noinline)) void pixel_write(const FastArgs *fap, float sx, float sy, float sz, int lofs, uint32_t segments,
                                                      uint32_t draws, uint32_t d, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, uint32_t v4) {
    const FastArgs &fa = *fap;
    const RenderArgs &a = fa.r;
    if (SUB) { /* a stratum row of a pixel: its colour sum goes to the virtual accumulator, the resolve kernel finishes the pixel */
        fa.vaccum[3 * (size_t)lofs] = sx; fa.vaccum[3 * (size_t)lofs + 1] = sy; fa.vaccum[3 * (size_t)lofs + 2] = sz;
        mort_rng_state st;
        st.d = d; st.v[0] = v0; st.v[1] = v1; st.v[2] = v2; st.v[3] = v3; st.v[4] = v4;
        st.boxmuller_flag = 0; st.boxmuller_flag_double = 0; st.boxmuller_extra = 0.f; st.boxmuller_extra_double = 0.;
        a.states[lofs] = st;
        const unsigned slot = 32u + 2u * (blockIdx.x & 31u);
        atomicAdd(&a.counters[slot], (unsigned long long)segments);
        atomicAdd(&a.counters[slot + 1u], (unsigned long long)draws);
        return;
    }
    V3 c = vscale(a.pixel_samples_scale, mk(sx, sy, sz)); /* camera.cuh:194-207 */
    if (c.x != c.x) c.x = 0.0f;
    if (c.y != c.y) c.y = 0.0f;
    if (c.z != c.z) c.z = 0.0f;
    if (fa.tile_cost) { /* the tile's key in the next frame's order: its LONGEST pixel (a frame with a few pixels per lane ends when its longest chains
                         * do, so those must start first, and a tile with one long pixel among 63 short ones has a small sum), or the tile's sum */
        const int lyl = lofs / a.width, xl = lofs - lyl * a.width;
        unsigned *tc = &fa.tile_cost[(lyl >> 3) * fa.tiles_x + (xl >> 3)];
        if (fa.tile_key_sum) atomicAdd(tc, segments); else atomicMax(tc, segments);
    }
    if (!PROBE) {
        if (a.accum) { a.accum[3 * lofs] = c.x; a.accum[3 * lofs + 1] = c.y; a.accum[3 * lofs + 2] = c.z; }
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
        a.rgba[lofs] = out;
        if (a.seg_px) a.seg_px[lofs] = segments;
        mort_rng_state st;
        st.d = d; st.v[0] = v0; st.v[1] = v1; st.v[2] = v2; st.v[3] = v3; st.v[4] = v4;
        st.boxmuller_flag = 0; st.boxmuller_flag_double = 0; st.boxmuller_extra = 0.f; st.boxmuller_extra_double = 0.;
        a.states[lofs] = st;
        /* per-pixel totals: 32 slots each, picked by workgroup (same-address atomics would queue up behind each other) */
        const unsigned slot = 32u + 2u * (blockIdx.x & 31u);
        atomicAdd(&a.counters[slot], (unsigned long long)segments);
        atomicAdd(&a.counters[slot + 1u], (unsigned long long)draws);
    }
}
struct PixelFetch { int got; /* 0, or 1 + the stratum row (SUB) */ int xy /* x | y << 16 | priority pixel << 31 */, lofs; uint32_t d, v0, v1, v2, v3, v4; };
/* one atomicAdd per wave per refill; the lanes that call this together take consecutive slots */
template <bool SUB = false>
__device__ __attribute__((noinline)) PixelFetch pixel_fetch(const FastArgs *fap, unsigned total_q, int role = -1, unsigned head_tiles = 0u) {
    const FastArgs &fa = *fap;
    const RenderArgs &a = fa.r;
    PixelFetch pf; pf.got = 0; pf.xy = 0; pf.lofs = 0; pf.d = pf.v0 = pf.v1 = pf.v2 = pf.v3 = pf.v4 = 0;
    const int vrows = SUB ? a.local_rows * fa.sub : a.local_rows; /* rows of the (virtual) image the tiles cover */
    bool done = false;
    while (!done) {
        unsigned long long need = __ballot(1);
        int cnt = __popcll(need);
        int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(need >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need, 0));
        unsigned q = 0;
        bool from_a = false;
        const unsigned total_a = SUB ? 0u : (role >= 0 ? head_tiles * 64u : (unsigned)fa.prio_tiles * 64u);
        bool by_role = false;
        if (role >= 0 && total_a) { /* heavy waves (role 1) take the head of the cost order first, the others (role 0) the rest first; either takes what is left of the other part */
            by_role = true;
            const bool head_first = role == 1;
            unsigned b1 = 0;
            if (rank == 0) b1 = atomicAdd(head_first ? fa.next_q + 1 : fa.next_q, (unsigned)cnt);
            b1 = __shfl(b1, __ffsll((long long)need) - 1);
            const unsigned q1 = b1 + (unsigned)rank;
            const bool got1 = head_first ? (q1 < total_a) : (q1 < total_q - total_a);
            if (got1) { q = head_first ? q1 : total_a + q1; from_a = head_first; }
            else {
                const unsigned long long need2 = __ballot(1);
                const int cnt2 = __popcll(need2);
                const int rank2 = __builtin_amdgcn_mbcnt_hi((unsigned)(need2 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need2, 0));
                unsigned b2 = 0;
                if (rank2 == 0) b2 = atomicAdd(head_first ? fa.next_q : fa.next_q + 1, (unsigned)cnt2);
                b2 = __shfl(b2, __ffsll((long long)need2) - 1);
                const unsigned q2 = b2 + (unsigned)rank2;
                const bool got2 = head_first ? (q2 < total_q - total_a) : (q2 < total_a);
                if (!got2) break; /* both parts are empty */
                q = head_first ? total_a + q2 : q2; from_a = !head_first;
            }
        } else if (total_a) { /* the head of the cost order: at most prio_lanes pixels of it per fetch */
            const int ka = cnt < fa.prio_lanes ? cnt : fa.prio_lanes;
            unsigned base_a = 0;
            if (rank == 0) base_a = atomicAdd(fa.next_q + 1, (unsigned)ka);
            base_a = __shfl(base_a, __ffsll((long long)need) - 1);
            if (rank < ka && base_a + (unsigned)rank < total_a) { q = base_a + (unsigned)rank; from_a = true; }
        }
        if (!from_a && !by_role) {
            need = __ballot(1);
            cnt = __popcll(need);
            rank = __builtin_amdgcn_mbcnt_hi((unsigned)(need >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need, 0));
            unsigned base = 0;
            if (rank == 0) base = atomicAdd(fa.next_q, (unsigned)cnt);
            base = __shfl(base, __ffsll((long long)need) - 1);
            q = total_a + base + (unsigned)rank;
        }
        if (q >= total_q) break; /* pool empty */
        /* slot -> (tile rank, pixel of the tile).  A lane's chain advances one segment per round of its
         * wave's state loop, and a round is slowest when all 64 lanes carry long chains; neighbouring
         * pixels, on the other hand, keep a wave's rays coherent.  So the 64 slots of one fetch are
         * 64 >> spread_shift groups of neighbouring pixels, spread evenly over a whole GENERATION of the
         * cost order (the gen_tiles tiles that the chip's lanes take at once).  Generation g, chunk w,
         * group u -> tile rank g*G + (w + u*S) mod G, pixel group u of that tile: a rotation per u,
         * hence a bijection between slots and pixels. */
        int tslot = (int)(q >> 6), within = (int)(q & 63u);
        if (fa.gen_tiles > 0 && !from_a && !by_role) { /* (heavy-wave launches hand out whole tiles in order) */
            const int g = tslot / fa.gen_tiles, w = tslot - g * fa.gen_tiles;
            const int left = fa.tiles_total - g * fa.gen_tiles;
            const int G = left < fa.gen_tiles ? left : fa.gen_tiles;
            const int groups = 64 >> fa.spread_shift, u = within >> fa.spread_shift;
            const int S = G >= groups ? G / groups : 1;
            tslot = g * fa.gen_tiles + (int)(((unsigned)w + (unsigned)u * (unsigned)S) % (unsigned)G);
        }
        const int tile = fa.tile_order ? (int)fa.tile_order[tslot] : tslot;
        const int tx = tile % fa.tiles_x, ty = tile / fa.tiles_x;
        const int qx = tx * 8 + (within & 7), qly = ty * 8 + (within >> 3);
        if (qx < a.width && qly < vrows) {
            const int rly = SUB ? qly / fa.sub : qly; /* the pixel's local row; SUB: qly = row * sub + stratum row */
            pf.xy = qx | (global_row(rly, a.rank, a.nranks, a.rows_per_block) << 16) | (from_a ? (int)0x80000000u : 0);
            pf.lofs = qx + qly * a.width;
            pf.got = SUB ? 1 + (qly - rly * fa.sub) : 1;
            done = true;
        }
    }
    if (pf.got) {
        const mort_rng_state st = a.states[pf.lofs];
        pf.d = st.d; pf.v0 = st.v[0]; pf.v1 = st.v[1]; pf.v2 = st.v[2]; pf.v3 = st.v[3]; pf.v4 = st.v[4];
    }
    return pf;
}

/* PROBE = true is the 1-sample cost probe (its own symbol, so profiles keep the frame kernel's durations apart) */
/* DRAIN = true adds the drain mode below (chain-bound partitions; it costs the throughput-bound frame 5 % in registers).  A chain-bound
 * launch is latency-bound -- the frame ends when its longest pixel chain does, at the pace of one wave -- so its smaller workgroups are
 * compiled for the waves per SIMD they bring themselves (256 threads: 1, 384 / 512: 2) and keep the whole state in registers: no spills,
 * whose scratch latency a lone wave cannot hide */
/* SUB = true: MORT_MODE_THROUGHPUT's launch over (pixel, stratum row) work items with their own streams (FastArgs.sub) -- labelled, never parity.
 * (A thin wrapper kernel around a shared body would keep the three-parameter symbol names, but it perturbs the frame kernel's register
 * allocation: 136 instead of 122 scratch instructions) */
template <int BLOCK, bool PROBE, bool DRAIN = false, bool SUB = false>
__global__ void __launch_bounds__(BLOCK, BLOCK == 1024 ? 4 : (DRAIN && BLOCK < 768) ? (BLOCK + 255) / 256 : MORT_MIN_WAVES) mega_bvh_kernel(const FastArgs MORT_BVH_ARG_NAME) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    /* The by-value parameter is never named: round 2's kernel took its address for the out-of-line helpers, so hipcc kept a private
     * copy of the argument struct per lane (592 B of scratch) and read wave-uniform fields from it inside the state loop.  The kernarg
     * segment is copied into LDS once, the loop's uniform values are pinned in SGPRs, the helpers get the segment's own address. */
    __shared__ CamView s_cam; /* dev_render.h: get_ray's camera fields, read from LDS */
#ifdef MORT_BVH_PRIVATE_ARGS /* round 2's form, kept as a measured variant: helpers get &fa, hipcc keeps a private copy */
    const FastArgs *const fap = &fa_byval;
    const FastArgs &s_fa = fa_byval;
#else
    __shared__ FastArgs s_fa;
    const FastArgs *const fap = (const FastArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    {
        const uint32_t *src = (const uint32_t *)fap;
        uint32_t *dst = (uint32_t *)&s_fa;
        for (uint32_t i = threadIdx.x; i < (uint32_t)(sizeof(FastArgs) / 4); i += BLOCK) dst[i] = src[i];
    }
#endif
    {
        const uint4 *src = (const uint4 *)fap->hot_src;
        uint4 *dst = (uint4 *)lds;
        const uint32_t n16 = fap->hot_bytes >> 4;
        for (uint32_t i = threadIdx.x; i < n16; i += BLOCK) dst[i] = src[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) cam_view_fill(s_cam, s_fa.r);
    __syncthreads();
    const FastArgs &L = s_fa;
    /* both record arrays as 16-byte pieces: a child reference is already a piece index (dev_scene.h) */
    static_assert(sizeof(DNode4) == 16 * MORT_NODE4_PIECES && sizeof(DLeaf2) == 16 * MORT_LEAF2_PIECES, "record strides");
    const float4 *nodes4 = (const float4 *)(lds + BU_U(L.off_nodes4));
    const float4 *leafrecs = (const float4 *)(lds + BU_U(L.off_leafrecs));
    /* traversal stack [level][thread] u16, addressed by the LDS byte offset of the lane's next free entry */
    const uint32_t ts_base = BU_U(L.off_tstack) + 2u * threadIdx.x;
#define TS_AT(off) (*(unsigned short *)(lds + (off)))
    const DLambert *lambert = (const DLambert *)(lds + BU_U(L.off_lambert));
    const DMetal *metal = (const DMetal *)(lds + BU_U(L.off_metal));
    const DDielectric *dielectric = (const DDielectric *)(lds + BU_U(L.off_diel));
    const DLambert *dlight = (const DLambert *)(lds + BU_U(L.off_dlight));
    const DLambert *isotropic = (const DLambert *)(lds + BU_U(L.off_iso));
    const DSolid *solid = (const DSolid *)(lds + BU_U(L.off_solid));
    const DChecker *checker = (const DChecker *)(lds + BU_U(L.off_checker));

    const int node_first = BU_I(L.node_first), node_end = node_first + BU_I(L.node_count);
    const int th_s = BU_I(L.th_s), th_l = BU_I(L.th_l), t_keep = BU_I(L.t_keep);
    const int sqrt_spp = BU_I(L.r.sqrt_spp), bounce_limit = BU_I(L.r.bounce_limit);
    const float bg_x = BU_F(L.r.background.x), bg_y = BU_F(L.r.background.y), bg_z = BU_F(L.r.background.z);
    const int spp = sqrt_spp * sqrt_spp;
    const unsigned total_q = (unsigned)BU_I(L.tiles_total) * 64u;

    /* per-lane state */
    int state = ((int)(threadIdx.x & 63u) < BU_I(L.lane_cap)) ? ST_S : ST_DONE, kind = K_NEWPIX;
    int xy = 0, lofs = 0; /* x | y << 16 */
    Rng rng; rng.d = rng.v0 = rng.v1 = rng.v2 = rng.v3 = rng.v4 = 0; rng.draws = 0;
    V3 pixel_color = mk(0, 0, 0);
    int s_ij = 0, iter = 0; /* stratum s_i | s_j << 16; the sample index is s_j * sqrt_spp + s_i */
    uint32_t segments = 0;
    Ray ray; ray.o = mk(0, 0, 0); ray.d = mk(0, 0, 1); ray.tm = 0;
    float ray_time0 = 0;
    OwnRay orr; orr.ix = orr.iy = orr.iz = 1; orr.mx = orr.my = orr.mz = 0; orr.band = 0; orr.invlen = 1;
    float ray_a = 1, closest = 0;
    int best = -1;         /* closest hit so far: leaf << 16 | (second sphere of the leaf) << 15 */
    uint32_t node = 0;     /* T: own-tree node; L: leaf record */
    uint32_t spa = ts_base; /* pending children: next free entry of the lane's traversal stack */
    int flags = 0;          /* FL_TIE / FL_REF */
    /* bounce-stack levels below the LDS part: [level - DL][lane of the launch] in HBM, one coalesced 1 KB row per wave and level, touched
     * only by paths deeper than the LDS part (a private array is scratch memory sized for the deepest path in every lane) */
    float4 *stack_deep = BU_P(L.deep) + ((size_t)blockIdx.x * BLOCK + threadIdx.x);
    const size_t deep_stride = (size_t)gridDim.x * BLOCK;
    /* bit i set: bounce level i is a dielectric scatter, whose entry (k = (1,1,1), 1/pdf = 1) unwinds as
     * final = 0 + 1*((1,1,1)*final) = 0 + final exactly -- such levels are neither stored nor loaded */
    unsigned long long ident_mask = 0ull;
    float4 *stack_lds = (float4 *)(lds + BU_U(L.off_stack));
    const int DL = BU_I(L.stack_lds_depth);
#ifdef MORT_PROFILE_STATES
    unsigned long long *const counters_p = fap->r.counters;
    unsigned long long prof[6] = {0, 0, 0, 0, 0, 0}; /* steps/lanes for T, L, S (wave-uniform) */
    unsigned long long prof_truns = 0;
    unsigned long long profc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; /* cycles in T, L, S, scheduler; S parts: shade, finish, newpix, setup */
    unsigned long long ps0 = 0, ps1;
#define PROFS0() do { ps0 = __builtin_readcyclecounter(); } while (0)
#define PROFS(i) do { ps1 = __builtin_readcyclecounter(); profc[i] += ps1 - ps0; ps0 = ps1; } while (0)
    unsigned long long pt0 = __builtin_readcyclecounter(), pt1;
    const unsigned long long prof_r0 = __builtin_amdgcn_s_memrealtime();
#define PROF(i, lanes) do { prof[2 * (i)] += 1; prof[2 * (i) + 1] += (unsigned long long)(lanes); } while (0)
    unsigned long long profb[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; /* S branches: steps entered, lanes: metal, dielectric, lambertian, finish, get_ray, unwind iterations */
#ifdef MORT_PROFILE_FINE /* cycles of the parts of the shade step instead of the branch counts (whose atomics distort the step's time) */
#define PROFB(i) do { } while (0)
    unsigned long long proff[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; /* verify, hit record, metal, dielectric, lambert texture, lambert scatter, light, stack store */
#define PROFF(i) do { ps1 = __builtin_readcyclecounter(); proff[i] += ps1 - ps0; profc[4] += ps1 - ps0; ps0 = ps1; } while (0)
#else
#define PROFF(i) do { } while (0)
#define PROFB(i) do { const unsigned long long m_ = __ballot(1); if ((int)(threadIdx.x & 63) == __ffsll((long long)m_) - 1) { \
        atomicAdd(&counters_p[18 + 2 * (i)], 1ull); atomicAdd(&counters_p[19 + 2 * (i)], (unsigned long long)__popcll(m_)); } } while (0)
#endif
#define PROFC(i) do { pt1 = __builtin_readcyclecounter(); profc[i] += pt1 - pt0; pt0 = pt1; } while (0)
#else
#define PROF(i, lanes) do { } while (0)
#define PROFB(i) do { } while (0)
#define PROFF(i) do { } while (0)
#define PROFC(i) do { } while (0)
#define PROFS0() do { } while (0)
#define PROFS(i) do { } while (0)
#endif

    /* Drain mode (wave-uniform).  Once the pixel pool is empty, throughput no longer matters to this wave,
     * only when its last pixel finishes.  The lane that is furthest behind (fewest samples done) then decides which
     * state runs next, so its chain advances at the pace of a wave that carries it alone; the other lanes advance
     * whenever they share its state. */
    int leader = -1;
    const bool drain_rounds = DRAIN && BU_I(L.drain_rounds) == 1;
    const bool live_thresholds = BU_I(L.drain_rounds) == 3;
    /* Loop shape: the scheduler and the two traversal states form an INNER loop in which only (state, node, spa, kind, closest, best, flags)
     * change; the shade step, which rewrites the whole per-lane state, is the outer loop's body.  As one flat loop hipcc gave every step
     * -- box steps included -- a round trip of some fifty register copies at the common join of the three branches */
    bool running = true;
    while (running) {
      int nS = 0;
      for (;;) {
        const unsigned long long mT = __ballot(state == ST_T);
        const unsigned long long mL = __ballot(state == ST_L);
        const unsigned long long mS = __ballot(state == ST_S);
        if ((mT | mL | mS) == 0ull) { running = false; break; }
        const int nT = __popcll(mT), nL = __popcll(mL);
        nS = __popcll(mS);
        int pick;
        int e_s = th_s, e_l = th_l; /* batch thresholds as shares of the wave's live lanes (mega_gen.hip) */
        if (live_thresholds) { const int live = nT + nL + nS; e_s = (th_s * live + 63) >> 6; e_l = (th_l * live + 63) >> 6; }
        if (nS >= e_s) pick = ST_S;
        else if (nL >= e_l) pick = ST_L;
        else if (nT > 0) pick = ST_T;
        else pick = (nL >= nS) ? ST_L : ST_S;
        if (DRAIN && leader >= 0) {
            if (drain_rounds) pick = nT > 0 ? ST_T : nL > 0 ? ST_L : ST_S; /* rounds: every live lane advances one segment per shade step (mega_gen.hip) */
            else { const int ls = __builtin_amdgcn_readlane(state, leader); if (ls <= ST_S) pick = ls; }
        }
        PROFC(3);

                REGION("T");
        if (pick != ST_T && pick != ST_L) break; /* ST_S: the shade step is the outer loop's body */
        if (pick == ST_T) {
            /* ---- own-tree steps: both child boxes of one node, near child next, far child pushed ---- */
            int keep;
#ifdef MORT_PROFILE_STATES
            prof_truns++;
#endif
            do {
#pragma unroll
                for (int rep = 0; rep < MORT_T_UNROLL; rep++) {
                PROF(0, __popcll(__ballot(state == ST_T)));
                if (state == ST_T) {
                    /* four child boxes per step.  The children that pass are ordered by entry distance (the upper 16 bits of te, a
                     * positive float, order as integers; the child reference rides in the lower 16): nearest next, the others
                     * pushed farthest first.  The order only decides how soon `closest` shrinks -- equal distances are settled by
                     * the reference's own walk (FL_TIE), never by the order of visits */
                    const float4 *np = nodes4 + node;
                    const float4 bx0 = np[0], bx1 = np[1], by0 = np[2], by1 = np[3], bz0 = np[4], bz1 = np[5], be = np[6];
                    const uint4 ch = ((const uint4 *)np)[7];
                    float t0, t1, t2, t3;
#ifdef MORT_BVH_PACKED_FMA
                    bool m0, m1, m2, m3;
                    own_prune2(v2f_t{bx0.x, bx0.y}, v2f_t{bx1.x, bx1.y}, v2f_t{by0.x, by0.y}, v2f_t{by1.x, by1.y}, v2f_t{bz0.x, bz0.y}, v2f_t{bz1.x, bz1.y}, v2f_t{be.x, be.y}, orr, closest, t0, t1, m0, m1);
                    own_prune2(v2f_t{bx0.z, bx0.w}, v2f_t{bx1.z, bx1.w}, v2f_t{by0.z, by0.w}, v2f_t{by1.z, by1.w}, v2f_t{bz0.z, bz0.w}, v2f_t{bz1.z, bz1.w}, v2f_t{be.z, be.w}, orr, closest, t2, t3, m2, m3);
                    m2 = m2 || ch.z == 0xffffu; m3 = m3 || ch.w == 0xffffu;
#else
                    const bool m0 = own_prune(bx0.x, bx1.x, by0.x, by1.x, bz0.x, bz1.x, be.x, orr, closest, t0);
                    const bool m1 = own_prune(bx0.y, bx1.y, by0.y, by1.y, bz0.y, bz1.y, be.y, orr, closest, t1);
                    const bool m2 = own_prune(bx0.z, bx1.z, by0.z, by1.z, bz0.z, bz1.z, be.z, orr, closest, t2) || ch.z == 0xffffu;
                    const bool m3 = own_prune(bx0.w, bx1.w, by0.w, by1.w, bz0.w, bz1.w, be.w, orr, closest, t3) || ch.w == 0xffffu;
#endif
                    uint32_t k0 = m0 ? 0xffffffffu : ((__float_as_uint(t0) & 0xffff0000u) | ch.x);
                    uint32_t k1 = m1 ? 0xffffffffu : ((__float_as_uint(t1) & 0xffff0000u) | ch.y);
                    uint32_t k2 = m2 ? 0xffffffffu : ((__float_as_uint(t2) & 0xffff0000u) | ch.z);
                    uint32_t k3 = m3 ? 0xffffffffu : ((__float_as_uint(t3) & 0xffff0000u) | ch.w);
#define MORT_CSWAP(a, b) do { const uint32_t lo_ = a < b ? a : b, hi_ = a < b ? b : a; a = lo_; b = hi_; } while (0)
                    MORT_CSWAP(k0, k1); MORT_CSWAP(k2, k3); MORT_CSWAP(k0, k2); MORT_CSWAP(k1, k3); MORT_CSWAP(k1, k2);
#undef MORT_CSWAP
                    if (k3 != 0xffffffffu) { TS_AT(spa) = (unsigned short)k3; spa += 2u * BLOCK; }
                    if (k2 != 0xffffffffu) { TS_AT(spa) = (unsigned short)k2; spa += 2u * BLOCK; }
                    if (k1 != 0xffffffffu) { TS_AT(spa) = (unsigned short)k1; spa += 2u * BLOCK; }
                    uint32_t next = k0;
                    const bool none = k0 == 0xffffffffu;
                    const bool have = !none || spa != ts_base;
                    if (none && spa != ts_base) { spa -= 2u * BLOCK; next = TS_AT(spa); }
                    if (!have) { state = ST_S; kind = K_SHADE; }
                    else { node = next & 0x7fffu; if (next & 0x8000u) state = ST_L; }
                }
                }
                keep = __popcll(__ballot(state == ST_T));
                if (DRAIN && leader >= 0) keep = drain_rounds ? (keep > 0 ? 64 : 0) : (__builtin_amdgcn_readlane(state, leader) == ST_T) ? 64 : 0;
                else if (keep > 0 && keep < t_keep) { /* few lanes left at nodes: hand control back only if another state has its batch together (mega_gen.hip) */
                    const int wL = __popcll(__ballot(state == ST_L)), wS = __popcll(__ballot(state == ST_S));
                    if (!(wS >= e_s || wL >= e_l)) keep = 64;
                }
            } while (keep >= t_keep);
            PROFC(0);
        }
        if (pick == ST_L) { /* not `else`: two simple diamonds, each merging a modified state with the unmodified one, leave hipcc nothing to copy at a common join */
                REGION("L");
            /* ---- leaf: sphere::hit on the one or two spheres of a reference leaf node (objects.cuh:60-77,690-692) ---- */
            PROF(1, nL);
            if (state == ST_L) {
                /* both spheres by value in one round trip to LDS (DLeaf2) */
                const float4 *lp = leafrecs + node;
                const float4 a0 = lp[0], a1 = lp[1], b0 = lp[2], b1 = lp[3];
                const uint32_t two = ((const uint4 *)lp)[5].z;
                DSphere sa, sb;
                sa.cx = a0.x; sa.cy = a0.y; sa.cz = a0.z; sa.radius = a0.w; sa.vx = a1.x; sa.vy = a1.y; sa.vz = a1.z; sa.mat = __float_as_uint(a1.w);
                sb.cx = b0.x; sb.cy = b0.y; sb.cz = b0.z; sb.radius = b0.w; sb.vx = b1.x; sb.vy = b1.y; sb.vz = b1.z; sb.mat = __float_as_uint(b1.w);
                const float ta = sphere_hit_root(sa, ray, ray_a, 0.001f, closest);
                if (ta != -1.0f) {
                    if (ta == closest && best >= 0) flags |= FL_TIE; /* the reference keeps whichever it visits last */
                    closest = ta; best = (int)(node << 16);
                }
                if (two == 2u) {
                    const float tb = sphere_hit_root(sb, ray, ray_a, 0.001f, closest);
                    if (tb != -1.0f) {
                        if (tb == closest && best >= 0) flags |= FL_TIE;
                        closest = tb; best = (int)((node << 16) | 0x8000u);
                    }
                }
                if (spa != ts_base) {
                    spa -= 2u * BLOCK;
                    const uint32_t next = TS_AT(spa);
                    node = next & 0x7fffu;
                    state = (next & 0x8000u) ? ST_L : ST_T;
                } else { state = ST_S; kind = K_SHADE; }
            }
            PROFC(1);
        }
      }
      if (!running) break;
        {
            /* ---- shade / finish / next sample / next pixel, then start the next ray ---- */
            PROF(2, nS);
            PROFS0();
            /* a path's result lives inside one shade step: set where the path ends, consumed by the finish part below.  (The two
             * degenerate launches -- no samples, no bounces -- enter a step with K_FINISH and a value of zero.) */
            V3 final_value = mk(0, 0, 0);
                REGION("S:verify");
            if (state == ST_S) {
                if (kind == K_SHADE) {
                    /* is the winner what bvh_node::hit returns?  (header comment) */
                    bool need_ref = flags != 0;
                    if (!need_ref && best >= 0) { /* the winner's leaf box, from its record */
                        const float4 *lp = leafrecs + (best >> 16);
                        const float4 b0 = lp[4], b1 = lp[5];
                        DBvhNode lb;
                        lb.xmin = b0.x; lb.xmax = b0.y; lb.ymin = b0.z; lb.ymax = b0.w; lb.zmin = b1.x; lb.zmax = b1.y; lb.skip = 0; lb.prims = 0;
                        need_ref = !slab_check(lb, ray, orr, closest);
                    }
                    if (need_ref) { /* rare (about one segment in 10^5): the reference's own walk */
                        atomicAdd(&fap->r.counters[3], 1ull);
                        const RefHit h = reference_walk(fap->r.sc.nodes, node_first, node_end, fap->r.sc.spheres, ray.o.x, ray.o.y, ray.o.z,
                                                        ray.d.x, ray.d.y, ray.d.z, ray.tm, ray_a);
                        best = h.best; closest = h.closest; /* an index into the scene's sphere table (HBM) */
                    }
                    PROFF(0);
                REGION("S:hitrecord");
                    if (best < 0) { /* camera.cuh:154-158 */
                        final_value = mk(bg_x, bg_y, bg_z);
                        kind = K_FINISH;
                    } else {
                        DSphere sp;
                        if (need_ref) sp = fap->r.sc.spheres[best];
                        else { /* the winner's copy in its leaf record */
                            const float4 *lp = leafrecs + (best >> 16) + ((best & 0x8000) ? 2 : 0);
                            const float4 s0 = lp[0], s1 = lp[1];
                            sp.cx = s0.x; sp.cy = s0.y; sp.cz = s0.z; sp.radius = s0.w; sp.vx = s1.x; sp.vy = s1.y; sp.vz = s1.z; sp.mat = __float_as_uint(s1.w);
                        }
                        const V3 p = ray_at(ray, closest);
                        const V3 outward = vdiv(vsub(p, sphere_center(sp, ray.tm)), sp.radius);
                        const bool front_face = vdot(ray.d, outward) < 0;
                        const V3 normal = front_face ? outward : vneg(outward);
                        const int mtype = DREF_TYPE(sp.mat), midx = DREF_IDX(sp.mat);
                        PROFF(1);
                        StackEntry e;
                        if (mtype == MORT_MAT_METAL) { /* materials.cuh:73-84 */
                REGION("S:metal");
                            PROFB(0);
                            const DMetal m = metal[midx];
                            V3 reflected = reflect(ray.d, normal);
                            reflected = vadd(vunit(reflected), vscale(m.fuzz, random_unit_vector(rng)));
                            ray.o = p; ray.d = reflected;
                            e.kx = 1.0f * m.r; e.ky = 1.0f * m.g; e.kz = 1.0f * m.b; e.rp = 1.0f;
                            PROFF(2);
                        } else if (mtype == MORT_MAT_DIELECTRIC) { /* materials.cuh:107-130 */
                REGION("S:dielectric");
                            PROFB(1);
                            const DDielectric m = dielectric[midx];
                            const float refraction_ratio = front_face ? m.inv_ior : m.ior;
                            const V3 unit_direction = vunit(ray.d);
                            const float cos_theta = (float)mort_fmin((double)vdot(vneg(unit_direction), normal), 1.0);
                            const float sin_theta = (float)mort_sqrt(1.0 - (double)(cos_theta * cos_theta));
                            const bool cant_refract = (double)(refraction_ratio * sin_theta) > 1.0;
                            V3 direction;
                            if (cant_refract || reflectance(cos_theta, refraction_ratio) > random_float(rng))
                                direction = reflect(unit_direction, normal);
                            else
                                direction = refract(unit_direction, normal, refraction_ratio);
                            ray.o = p; ray.d = direction;
                            e.kx = 1.0f; e.ky = 1.0f; e.kz = 1.0f; e.rp = 1.0f;
                            ident_mask |= (1ull << iter);
                            PROFF(3);
                        } else if (mtype == MORT_MAT_LAMBERTIAN || mtype == MORT_MAT_ISOTROPIC) {
                REGION("S:lambertian");
                            PROFB(2);
                            const bool lamb = (mtype == MORT_MAT_LAMBERTIAN);
                            const DLambert m = lamb ? lambert[midx] : isotropic[midx];
                            V3 attenuation = mk(m.r, m.g, m.b);
                            if (m.tex != 0) {
                                uint32_t tex = m.tex;
                                bool resolved = false;
                                for (int guard = 0; guard < 8 && !resolved; guard++) {
                                    const int tt = DREF_TYPE(tex), ti = DREF_IDX(tex);
                                    if (tt == MORT_TEXTURE_SOLID) { const DSolid sc = solid[ti]; attenuation = mk(sc.r, sc.g, sc.b); resolved = true; }
                                    else if (tt == MORT_TEXTURE_CHECKER) {
                                        const DChecker c = checker[ti];
                                        const int xi = mort_f2i(mort_floorf(c.inv_scale * p.x));
                                        const int yi = mort_f2i(mort_floorf(c.inv_scale * p.y));
                                        const int zi = mort_f2i(mort_floorf(c.inv_scale * p.z));
                                        tex = ((xi + yi + zi) % 2 == 0) ? c.even : c.odd;
                                    } else break;
                                }
                                if (!resolved) { /* image / noise / error pattern: tables stay in HBM */
                                    const V3Ret c = texture_value_uv(&fap->r.sc, tex, outward.x, outward.y, outward.z, p.x, p.y, p.z);
                                    attenuation = mk(c.x, c.y, c.z);
                                }
                            }
                            PROFF(4);
                REGION("S:lamb_scatter");
                            V3 dir;
                            float mat_pdf, scattering_pdf;
                            if (lamb) {
                                const Onb uvw = onb_from_w(normal);
                                dir = onb_local(uvw, random_cosine_direction(rng));
                                const V3 ud = vunit(dir);
                                const float cosine_theta = vdot(ud, uvw.w);
                                mat_pdf = mort_fmaxf(0, (float)((double)cosine_theta / 3.1415926));
                                const float cos_theta = vdot(normal, ud);
                                scattering_pdf = (cos_theta < 0) ? 0.0f : (float)((double)cos_theta / 3.141592565);
                            } else {
                                dir = random_unit_vector(rng);
                                mat_pdf = (float)(1 / (4 * 3.1415926));
                                scattering_pdf = (float)(1 / (4 * 3.1415926));
                            }
                            ray.o = p; ray.d = dir; ray.tm = ray_time0;
                            e.kx = scattering_pdf * attenuation.x; e.ky = scattering_pdf * attenuation.y; e.kz = scattering_pdf * attenuation.z;
                            e.rp = 1 / mat_pdf;
                            PROFF(5);
                        } else { /* diffuse_light or unknown tag: no scatter (materials.cuh:151-163) */
                REGION("S:light");
                            V3 emission = mk(0, 0, 0);
                            if (mtype == MORT_MAT_DIFFUSE_LIGHT && front_face) {
                                const DLambert m = dlight[midx];
                                if (m.tex == 0) emission = mk(m.r, m.g, m.b);
                                else { const V3Ret c = texture_value_uv(&fap->r.sc, m.tex, outward.x, outward.y, outward.z, p.x, p.y, p.z); emission = mk(c.x, c.y, c.z); }
                            }
                            final_value = emission;
                            kind = K_FINISH;
                            PROFF(6);
                        }
                REGION("S:stackstore");
                        if (kind == K_SHADE) {
                            if (!((ident_mask >> iter) & 1ull)) {
                                float4 e4; e4.x = e.kx; e4.y = e.ky; e4.z = e.kz; e4.w = e.rp;
                                if (iter < DL) stack_lds[iter * BLOCK + threadIdx.x] = e4;
                                else stack_deep[(size_t)(iter - DL) * deep_stride] = e4;
                            }
                            iter++;
                            if (iter >= bounce_limit) { final_value = mk(0, 0, 0); kind = K_FINISH; } /* camera.cuh:161-163 */
                        }
                    }
                }
                REGION("S:finish");
                PROFS(4);
                if (kind == K_FINISH) { /* unwind + accumulate (camera.cuh:165-173,190) */
                    PROFB(3);
                    /* an identity (dielectric) level is final = 0 + final: it only turns -0 into +0, and every level
                     * leaves a value without -0, so only an identity level that comes FIRST (deepest) can matter;
                     * the loop visits the stored levels alone, deepest first */
                    if (iter > 0) {
                        unsigned long long todo = ~ident_mask & (iter >= 64 ? ~0ull : ((1ull << iter) - 1ull));
                        if ((ident_mask >> (iter - 1)) & 1ull) final_value = vadd(mk(0, 0, 0), final_value);
                        while (todo != 0ull) {
                            const int lvl = 63 - __builtin_clzll(todo);
                            todo &= ~(1ull << lvl);
                            StackEntry e;
                            const float4 e4 = (lvl < DL) ? stack_lds[lvl * BLOCK + threadIdx.x] : stack_deep[(size_t)(lvl - DL) * deep_stride];
                            e.kx = e4.x; e.ky = e4.y; e.kz = e4.z; e.rp = e4.w;
                            const V3 t = vmul(mk(e.kx, e.ky, e.kz), final_value);
                            final_value = vadd(mk(0, 0, 0), vscale(e.rp, t));
                        }
                        iter = 0;
                    }
                    ident_mask = 0ull;
                    pixel_color = vadd(pixel_color, final_value);
                    s_ij++;
                    bool more; /* samples left in this work item (SUB: one stratum row) */
                    if constexpr (SUB) {
                        more = (s_ij & 0xffff) != sqrt_spp;
                    } else {
                        if ((s_ij & 0xffff) == sqrt_spp) s_ij = (s_ij & ~0xffff) + 0x10000;
                        more = (s_ij >> 16) < sqrt_spp;
                    }
                    if (more) {
                        kind = K_NEWSAMPLE;
                    } else {
                        pixel_write<PROBE, SUB>(fap, pixel_color.x, pixel_color.y, pixel_color.z, lofs, segments, rng.draws, rng.d, rng.v0, rng.v1, rng.v2, rng.v3, rng.v4);
                        kind = K_NEWPIX;
                    }
                }
                REGION("S:newpix");
                PROFS(5);
                if (kind == K_NEWPIX) {
                    const PixelFetch pf = pixel_fetch<SUB>(fap, total_q);
                    if (!pf.got) state = ST_DONE;
                    else {
                        xy = pf.xy; lofs = pf.lofs;
                        rng.d = pf.d; rng.v0 = pf.v0; rng.v1 = pf.v1; rng.v2 = pf.v2; rng.v3 = pf.v3; rng.v4 = pf.v4;
                        rng.draws = 0;
                        pixel_color = mk(0, 0, 0);
                        s_ij = SUB ? ((pf.got - 1) << 16) : 0; segments = 0;
                        kind = (spp > 0) ? K_NEWSAMPLE : K_FINISH;
                        if (spp <= 0) { /* degenerate: zero samples -> 0 * inf = NaN -> 0 */
                            final_value = mk(0, 0, 0);
                        }
                    }
                }
                REGION("S:getray");
                PROFS(6);
                if (state != ST_DONE) {
                    if (kind == K_NEWSAMPLE) { /* camera.cuh:187-190 */
                        PROFB(4);
                        ray = get_ray(s_cam, xy & 0xffff, (xy >> 16) & 0x7fff, rng, s_ij & 0xffff, s_ij >> 16);
                        ray_time0 = ray.tm;
                        iter = 0;
                        kind = K_SHADE;
                        if (bounce_limit <= 0) { final_value = mk(0, 0, 0); kind = K_FINISH; }
                    }
                REGION("S:setup");
                    if (kind == K_SHADE) { /* start world::hit for the new ray */
                        ray_a = vlen2(ray.d);
                        orr.ix = 1.0f / ray.d.x; orr.iy = 1.0f / ray.d.y; orr.iz = 1.0f / ray.d.z;
                        orr.mx = ray.o.x * orr.ix; orr.my = ray.o.y * orr.iy; orr.mz = ray.o.z * orr.iz;
                        const float mm = __builtin_fmaxf(__builtin_fmaxf(mort_fabsf(orr.mx), mort_fabsf(orr.my)), mort_fabsf(orr.mz));
                        orr.band = mm * 4.76837158203125e-07f; /* 2^-21 */
                        orr.invlen = 1.01f / mort_sqrtf(ray_a);
                        const bool ordinary = own_inv_ok(orr.ix) && own_inv_ok(orr.iy) && own_inv_ok(orr.iz) && (mm < 1e30f);
                        closest = __builtin_inff();
                        best = -1;
                        node = 0; spa = ts_base;
                        flags = ordinary ? 0 : FL_REF;
                        segments++;
                        state = ordinary ? ST_T : ST_S;
                    }
                    /* kind == K_FINISH here (bounce_limit 0 or spp 0): stays in ST_S for the next S step */
                }
            }
                REGION("S:end");
            /* pool empty (some lane found no pixel): follow the lane with the fewest samples done */
            if (DRAIN && __ballot(state == ST_DONE) != 0ull) {
                unsigned key = (state <= ST_S) ? ((unsigned)((s_ij >> 16) * sqrt_spp + (s_ij & 0xffff)) << 6) | (threadIdx.x & 63u) : 0xffffffffu;
                for (int off = 32; off > 0; off >>= 1) { const unsigned o = (unsigned)__shfl_xor((int)key, off); key = o < key ? o : key; }
                leader = (key == 0xffffffffu) ? -1 : __builtin_amdgcn_readfirstlane((int)(key & 63u));
            }
            PROFS(7);
            PROFC(2);
        }
    }
    if ((threadIdx.x & 63) == 0) {
#ifdef MORT_PROFILE_STATES
#ifdef MORT_PROFILE_FINE
        for (int k = 0; k < 12; k++) atomicAdd(&counters_p[18 + k], proff[k]);
#endif
        for (int k = 0; k < 6; k++) atomicAdd(&counters_p[4 + k], prof[k]);
        atomicAdd(&counters_p[30], prof_truns);
        for (int k = 0; k < 4; k++) atomicAdd(&counters_p[10 + k], profc[k]);
        for (int k = 4; k < 8; k++) atomicAdd(&counters_p[14 + k - 4], profc[k]);
        /* MORT_WAVE_LINES=1: one record per wave (same columns as mega_gen.hip's, the M columns empty) */
        if (fap->wave_log) {
            unsigned long long *w = fap->wave_log + 16 * ((size_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6));
            w[0] = blockIdx.x; w[1] = threadIdx.x >> 6; w[2] = __builtin_amdgcn_s_memrealtime() - prof_r0;
            w[3] = prof[0]; w[4] = prof[2]; w[5] = 0; w[6] = prof[4]; w[7] = profc[0]; w[8] = profc[1]; w[9] = 0; w[10] = profc[2]; w[11] = profc[3];
        }
        (void)profb;
#endif
    }
}

#endif
