/*
 * mega_gen.hip -- state-machine megakernel for worlds without reference BVHs: reference scenes 2..9 (two spheres,
 * earth, Perlin, quads, Cornell box, Cornell smoke, the book-2 final scene), i.e. everything the reference traces
 * with the brute-force loops of world::hit (world.cuh:122-168) over spheres, quads, translate / rotate_y instances,
 * hittable lists and constant media (objects.cuh:60-77,190-215,268-278,334-366,396-434,471-486), shaded with all five
 * materials, four textures and the light / material mixture pdf (materials.cuh, textures.cuh, pdf.cuh).
 *
 * CDNA4 design (the generalisation of mega_bvh.h to arbitrary object graphs):
 *
 *  - Unified tree in LDS.  All solid primitives of the world -- whatever instance chain or list they sit under --
 *    are leaves of ONE tree of this build (scene_compile.h build_unified: SAH, 64-byte two-child nodes, leaves of at
 *    most MORT_GEN_LEAF_MAX primitives).  Nodes, leaf table, primitive entries, transform chains, item / material /
 *    texture tables are copied to LDS once per workgroup; spheres and quads too when they fit (scenes 2..7), else
 *    they stay L2-resident (final scene: 2401 quads x 80 B).  A box step reads one node with four ds_read_b128.
 *    The reference scans every primitive per ray (3 408 tests per segment in the final scene).
 *
 *  - Why the tree gives the scan's answer: header of build_unified.  The scan's result is the primitive with the
 *    smallest own t, ties -> scanned last; the walk sees every primitive whose own t is <= the final closest, keeps
 *    the minimum (equal t: the higher scan rank, as the scan would); rays whose reciprocal direction is not an ordinary float
 *    repeat the segment with the scan itself (scan_solids, out of line, about never); equal t is resolved in place by scan rank.  Constant media are evaluated
 *    after the tree with the final closest_so_far, in scan order, as world::hit does (their RNG draw depends on it).
 *
 *  - Per-lane state machine, wave-level scheduling (as mega_bvh.h): T both child boxes of a node, L the primitives
 *    of a leaf, M the constant media, S shade / finish / next sample / next pixel.  Each wave iteration runs one
 *    state's code for the lanes in it, picked from ballot / popcount thresholds; a lane whose path ends never waits
 *    for the wave's longest path, and lanes fetch pixels from one atomic counter (cost-ordered 8x8 tiles).
 *
 *  - XORWOW state in 6 VGPRs for the pixel's lifetime; bounce stack (scattering_pdf * attenuation, 1 / pdf) with its
 *    first levels in LDS ([depth][thread], conflict-free b128), identity (dielectric) levels not stored at all.
 *
 * Shading is dev_shade.h's shade_hit(), the same body the one-lane-per-pixel kernel and the host loop run.
 */
#include <hip/hip_runtime.h>

#include "mega_gen.h"
#include "dev_gen.h"
#include "dev_shade.h"

#pragma clang fp contract(off)

enum { G_T = 0, G_L = 1, G_M = 2, G_S = 3, G_DONE = 4 };

#ifndef MORT_GEN_MIN_WAVES
#define MORT_GEN_MIN_WAVES 3
#endif

/* the scan for the rays the walk does not decide (equal t, non-ordinary reciprocals): out of line, reads the HBM copy */
struct ScanHit { uint32_t best; float closest; };
__device__ __attribute__((noinline)) ScanHit scan_solids(const DScene *scp, int first_medium, const int *chains, int n_chains,
                                                         float ox, float oy, float oz, float dx, float dy, float dz, float tm) {
    Ray r; r.o = mk(ox, oy, oz); r.d = mk(dx, dy, dz); r.tm = tm;
    ScanHit h;
    gen_scan_solids(*scp, first_medium, chains, n_chains, r, h.closest, h.best);
    return h;
}

/* ---- launch arguments without private memory ----
 * Round 2's kernel took `&ga` for its out-of-line helpers, so hipcc kept a PRIVATE copy of the 800-byte argument struct per lane
 * and read wave-uniform fields from it inside the state loop: 38 KB of scratch per wave for the copy alone, far more than the L2
 * share of a CU, so every such read was a trip to the Infinity Cache (1.0-1.5 TB of L2 <-> fabric traffic per frame of the final
 * scene).  Now: the by-value parameter is never named; the kernarg segment is copied once into LDS (s_ga), every
 * wave-uniform value the state loop needs is moved from there into an SGPR (readfirstlane: the compiler can neither re-load nor
 * re-materialise it), and the rare out-of-line helpers get the kernarg segment's own address. */
#ifndef MORT_GEN_WAVES_512
#define MORT_GEN_WAVES_512 2
#endif
/* SUB = true: MORT_MODE_THROUGHPUT's launch over (pixel, stratum row) work items with their own streams (mega_bvh.h FastArgs.sub) -- labelled, never parity */
template <int BLOCK, bool PRIMS_LDS, bool SUB = false>
__global__ void __launch_bounds__(BLOCK, (BLOCK == 1024 ? 4 : BLOCK == 512 ? MORT_GEN_WAVES_512 : MORT_GEN_MIN_WAVES)) mega_gen_kernel(const GenArgs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ GenArgs s_ga;  /* the launch arguments, read from LDS (never from a private copy) */
    __shared__ CamView s_cam; /* dev_render.h: get_ray's camera fields */
    const GenArgs *const gap = (const GenArgs *)__builtin_amdgcn_kernarg_segment_ptr(); /* for the out-of-line helpers */
    {
        const uint32_t *src = (const uint32_t *)gap;
        uint32_t *dst = (uint32_t *)&s_ga;
        for (uint32_t i = threadIdx.x; i < (uint32_t)(sizeof(GenArgs) / 4); i += BLOCK) dst[i] = src[i];
    }
    {
        const uint4 *src = (const uint4 *)gap->f.hot_src;
        uint4 *dst = (uint4 *)lds;
        const uint32_t n16 = gap->f.hot_bytes >> 4;
        for (uint32_t i = threadIdx.x; i < n16; i += BLOCK) dst[i] = src[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) cam_view_fill(s_cam, s_ga.f.r);
    __syncthreads();
    const GenArgs &L = s_ga;
    /* ---- wave-uniform values of the state loop, pinned in SGPRs ---- */
    const DNodeQ *nodes2 = (const DNodeQ *)(lds + uni_u(L.o_nodes));
    const uint32_t *entries = (const uint32_t *)(lds + uni_u(L.o_entries));
    const int *chains = (const int *)(lds + uni_u(L.o_chains));
    unsigned short *tstack = (unsigned short *)(lds + uni_u(L.f.off_tstack)) + threadIdx.x; /* [level * BLOCK] */
    /* this kernel's view of the scene: tables in LDS, big / rare ones (texels, Perlin tables, image descriptors for the
     * out-of-line lookups; primitives of big worlds) in HBM */
    DScene lsc;
    lsc.n_items = uni_i(L.f.r.sc.n_items); lsc.n_subitems = 0; lsc.nodes = nullptr; lsc.n_nodes = 0;
    lsc.n_spheres = 0; lsc.n_quads = 0; lsc.n_xforms = 0; lsc.n_media = 0; lsc.blob_bytes = 0; lsc.lds_bytes = 0;
    lsc.items = (const DItem *)(lds + uni_u(L.o_items)); lsc.subitems = (const DItem *)(lds + uni_u(L.o_subitems));
    lsc.xforms = (const DXform *)(lds + uni_u(L.o_xforms)); lsc.neg_inv_density = (const double *)(lds + uni_u(L.o_media));
    lsc.lambert = (const DLambert *)(lds + uni_u(L.o_lambert)); lsc.metal = (const DMetal *)(lds + uni_u(L.o_metal));
    lsc.dielectric = (const DDielectric *)(lds + uni_u(L.o_diel)); lsc.dlight = (const DLambert *)(lds + uni_u(L.o_dlight));
    lsc.isotropic = (const DLambert *)(lds + uni_u(L.o_iso)); lsc.solid = (const DSolid *)(lds + uni_u(L.o_solid));
    lsc.checker = (const DChecker *)(lds + uni_u(L.o_checker)); lsc.image = (const DImage *)(lds + uni_u(L.o_image));
    lsc.list_first = (const int *)(lds + uni_u(L.o_lfirst)); lsc.list_count = (const int *)(lds + uni_u(L.o_lcount));
    lsc.image_hbm = uni_p(L.f.r.sc.image_hbm); lsc.texels = uni_p(L.f.r.sc.texels); lsc.noise = uni_p(L.f.r.sc.noise);
    if (PRIMS_LDS) {
        lsc.spheres = (const DSphere *)(lds + uni_u(L.o_spheres)); lsc.quads = (const DQuad *)(lds + uni_u(L.o_quads));
        lsc.wspheres = (const DSphere *)(lds + uni_u(L.o_wspheres)); lsc.wquads = (const DQuad *)(lds + uni_u(L.o_wquads));
        lsc.list_types = (const int *)(lds + uni_u(L.o_ltypes)); lsc.list_idxs = (const int *)(lds + uni_u(L.o_lidxs));
    } else {
        lsc.spheres = uni_p(L.f.r.sc.spheres); lsc.quads = uni_p(L.f.r.sc.quads);
        lsc.wspheres = uni_p(L.f.r.sc.wspheres); lsc.wquads = uni_p(L.f.r.sc.wquads);
        lsc.list_types = uni_p(L.f.r.sc.list_types); lsc.list_idxs = uni_p(L.f.r.sc.list_idxs);
    }
    const DSphere *spheres = lsc.spheres;
    const DQuad *quads = lsc.quads;
    const uint32_t *ranks = uni_p(L.ranks);
    const int n_spheres = uni_i(L.n_spheres);

    const int th_s = uni_i(L.f.th_s), th_l = uni_i(L.f.th_l), t_keep = uni_i(L.f.t_keep), th_m = uni_i(L.th_m);
    const int sqrt_spp = uni_i(L.f.r.sqrt_spp), bounce_limit = uni_i(L.f.r.bounce_limit);
    const int light_type = uni_i(L.f.r.light_type), light_idx = uni_i(L.f.r.light_idx);
    const float bg_x = uni_f(L.f.r.background.x), bg_y = uni_f(L.f.r.background.y), bg_z = uni_f(L.f.r.background.z);
    const int spp = sqrt_spp * sqrt_spp;
    const unsigned total_q = (unsigned)uni_i(L.f.tiles_total) * 64u;
    const int n_items = lsc.n_items, first_medium = uni_i(L.first_medium);
    const bool has_media = first_medium < n_items;
    const uint32_t root = uni_u(L.root);
    const float g_x = uni_f(L.gx), g_y = uni_f(L.gy), g_z = uni_f(L.gz), g_R = uni_f(L.gR), g_mnear = uni_f(L.mnear), g_kmin = uni_f(L.kmin);
    const int probe = uni_i(L.probe);
    const int DL = uni_i(L.f.stack_lds_depth);
    float4 *stack_lds = (float4 *)(lds + uni_u(L.f.off_stack)) + threadIdx.x; /* [level * BLOCK] */
    /* bounce-stack levels below the LDS part: [level - DL][lane of the launch] in HBM, one coalesced 1 KB row per wave and level
     * (a private array would be scratch memory, sized for the worst case in every lane) */
    float4 *stack_deep = uni_p(L.f.deep) + ((size_t)blockIdx.x * BLOCK + threadIdx.x);
    const size_t deep_stride = (size_t)gridDim.x * BLOCK;

    /* per-lane state */
    /* heavy waves (FastArgs.heavy_*): a few lanes per wave on the head of the cost order */
    unsigned head_tiles = 0u;
    int role = -1;
    if (!SUB && uni_i(L.f.heavy_mod) > 0 && uni_p(L.f.prio_dev) != nullptr) {
        head_tiles = uni_u(*uni_p(L.f.prio_dev));
        const int wave_global = (int)(blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6));
        role = uni_i((wave_global % uni_i(L.f.heavy_mod)) < uni_i(L.f.heavy_num) ? 1 : 0); /* wave-uniform: into an SGPR */
        if (head_tiles == 0u) role = -1;
    }
    const int my_cap = role == 1 ? uni_i(L.f.heavy_cap) : uni_i(L.f.lane_cap);
    int state = ((int)(threadIdx.x & 63u) < my_cap) ? G_S : G_DONE, kind = K_NEWPIX; /* FastArgs.lane_cap: 64 = every lane takes pixels */
    int xy = 0, lofs = 0;
    Rng rng; rng.d = rng.v0 = rng.v1 = rng.v2 = rng.v3 = rng.v4 = 0; rng.draws = 0;
    V3 pixel_color = mk(0, 0, 0);
    int s_ij = 0, iter = 0;
    uint32_t segments = 0;
    Ray ray; ray.o = mk(0, 0, 0); ray.d = mk(0, 0, 1); ray.tm = 0;
    float ray_time0 = 0;
    GenRay gr; gr.ix = gr.iy = gr.iz = 1; gr.mx = gr.my = gr.mz = 0; gr.band = 0;
    float ray_a = 1, closest = 0;
    uint32_t best = GBEST_NONE; /* entry code of the closest hit so far */
    uint32_t node = 0;          /* T: tree node; L: leaf */
    int sp = 0, flags = 0;      /* pending far children; GFL_REF */
    unsigned long long ident_mask = 0ull;

#ifdef MORT_PROFILE_STATES
    unsigned long long gp_steps[4] = {0, 0, 0, 0}, gp_lanes[4] = {0, 0, 0, 0}, gp_cyc[5] = {0, 0, 0, 0, 0}, gp_lprims = 0, gp_liters = 0;
    unsigned long long gp_sp[6] = {0, 0, 0, 0, 0, 0}, gps0 = 0, gps1; /* S parts: stack store, finish, new pixel, new ray, decode, shade */
    unsigned long long gpt0 = __builtin_readcyclecounter(), gpt1;
    const unsigned long long gp_r0 = __builtin_amdgcn_s_memrealtime();
#define GPROFS0() do { gps0 = __builtin_readcyclecounter(); } while (0)
#define GPROFS(i) do { gps1 = __builtin_readcyclecounter(); gp_sp[i] += gps1 - gps0; gps0 = gps1; } while (0)
#define GPROF(i, lanes) do { gp_steps[i] += 1; gp_lanes[i] += (unsigned long long)(lanes); } while (0)
#define GPROFC(i) do { gpt1 = __builtin_readcyclecounter(); gp_cyc[i] += gpt1 - gpt0; gpt0 = gpt1; } while (0)
#else
#define GPROF(i, lanes) do { } while (0)
#define GPROFC(i) do { } while (0)
#define GPROFS0() do { } while (0)
#define GPROFS(i) do { } while (0)
#endif
    /* Loop shape: the scheduler and the three search states (T, L, M) form an INNER loop in which only the search's own variables (state,
     * node, sp, kind, closest, best, flags; M: the stream) change; the shade step, which rewrites the whole per-lane state, is the outer
     * loop's body.  As one flat loop hipcc gave every step -- box steps included -- a round trip of some fifty register copies at the
     * common join of the four branches (mega_bvh.h) */
    bool running = true;
    bool draining = false; /* wave-uniform: some lane of this wave found the pixel pool empty */
    const int drain_mode = uni_i(L.drain_mode);
    while (running) {
      int nS = 0;
      for (;;) {
        const unsigned long long mT = __ballot(state == G_T);
        const unsigned long long mL = __ballot(state == G_L);
        const unsigned long long mM = __ballot(state == G_M);
        const unsigned long long mS = __ballot(state == G_S);
        if ((mT | mL | mM | mS) == 0ull) { running = false; break; }
        const int nT = __popcll(mT), nL = __popcll(mL), nM = __popcll(mM);
        nS = __popcll(mS);
        int pick;
        /* the batch thresholds are shares of the wave's LIVE lanes (drain_mode 3): once the pool is empty and lanes retire, fixed counts
         * are never reached and the steps degenerate to whoever happens to wait (the frame's last waves ran 1.6 shade steps per segment) */
        int e_s = th_s, e_m = th_m, e_l = th_l;
        if (drain_mode == 3) { const int live = nT + nL + nM + nS; e_s = (th_s * live + 63) >> 6; e_m = (th_m * live + 63) >> 6; e_l = (th_l * live + 63) >> 6; }
        if (nS >= e_s) pick = G_S;
        else if (nM >= e_m) pick = G_M;
        else if (nL >= e_l) pick = G_L;
        else if (nT > 0) pick = G_T;
        else pick = (nL >= nS && nL >= nM) ? G_L : (nM >= nS ? G_M : G_S);
        /* Once the pool is empty (draining, wave-uniform) throughput no longer matters to this wave, only when its longest chain ends, and a
         * chain advances one segment per shade step it takes part in.  With the throughput thresholds the last lanes of a wave drift out of
         * phase (the frame's last waves ran 1.6 shade steps per segment of their longest pixel): from here on the wave runs in ROUNDS -- box
         * steps until no lane is at a node, leaf steps until none is at a leaf, media, and the shade step only when every live lane waits for
         * it -- so every live lane advances one segment per round.
         * drain_mode 1 (measured, not the default): follow ONE lane (priority pixels, FastArgs.prio_lanes) instead. */
        int leader = -1;
        if (draining) {
            if (drain_mode == 1) {
                const unsigned long long mP = __ballot(xy < 0 && state != G_DONE);
                leader = mP != 0ull ? __ffsll((long long)mP) - 1 : -1;
                if (leader >= 0) pick = __builtin_amdgcn_readlane(state, leader);
            } else if (drain_mode == 2) {
                pick = nT > 0 ? G_T : nL > 0 ? G_L : nM > 0 ? G_M : G_S;
            }
        } else if (drain_mode == 1) {
            const unsigned long long mP = __ballot(xy < 0 && state != G_DONE);
            leader = mP != 0ull ? __ffsll((long long)mP) - 1 : -1;
            if (leader >= 0) pick = __builtin_amdgcn_readlane(state, leader);
        }
        GPROFC(4);

        if (pick != G_T && pick != G_L && pick != G_M) break; /* G_S: the shade step is the outer loop's body */
        /* three sequential diamonds, not an if / else-if chain: each merges a modified search state with the unmodified one, so hipcc has nothing to
         * copy at a common join (mega_bvh.h: seven register copies per scheduler pass otherwise) */
        if (pick == G_T) {
            /* ---- box steps: both child boxes of one node, near child next, far child pushed ---- */
            int keep;
            do {
#pragma unroll
                for (int rep = 0; rep < MORT_T_UNROLL; rep++) {
                    GPROF(0, __popcll(__ballot(state == G_T)));
                    if (state == G_T) {
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
                        const bool have = !none || sp > 0;
                        if (none && sp > 0) { sp--; next = tstack[sp * BLOCK]; }
                        if (!have) { state = has_media ? G_M : G_S; kind = K_SHADE; }
                        else { node = next & 0x7fffu; if (next & 0x8000u) state = G_L; }
                    }
                }
                keep = __popcll(__ballot(state == G_T));
                if (leader >= 0) keep = (__builtin_amdgcn_readlane(state, leader) == G_T) ? 64 : 0;
                else if (draining && drain_mode == 2) keep = keep > 0 ? 64 : 0; /* rounds: until no lane is at a node */
                else if (keep > 0 && keep < t_keep) {
                    /* few lanes are left at nodes: hand control back only if another state has its batch together -- otherwise the scheduler
                     * would pick the box steps again (a fifth of the cycles of the frame's last waves went into such passes) */
                    const int wL = __popcll(__ballot(state == G_L)), wM = __popcll(__ballot(state == G_M)), wS = __popcll(__ballot(state == G_S));
                    if (!(wS >= e_s || wM >= e_m || wL >= e_l)) keep = 64;
                }
            } while (keep >= t_keep);
            GPROFC(0);
        }
        if (pick == G_L) {
            GPROF(1, nL);
            /* ---- leaf: the own hit test of each primitive of the leaf, in its own frame
             *      (sphere::hit objects.cuh:60-77, quad::hit :190-215 under translate / rotate_y :268-278,334-366) ---- */
            uint32_t pos = 0;
            int cnt = 0;
            if (state == G_L) { pos = GLEAF_FIRST(node); cnt = (int)GLEAF_COUNT(node); } /* a leaf reference is a run of entries */
            /* the record of the NEXT primitive is requested before this one is tested (with the primitives in HBM / L2 -- the final scene's
             * 2 401 quads do not fit in LDS -- a leaf is otherwise a chain of dependent load -> test -> load); two primitives per trip through
             * two sets of registers, so that nothing is copied from "next" to "current" */
            uint32_t e0 = 0, e1 = 0;
            PrimRec r0, r1;
            { const v4f_t z = {0.f, 0.f, 0.f, 0.f}; r0.a = r0.b = r0.c = r0.d = r0.e = z; r1 = r0; }
            if (cnt > 0) { e0 = entries[pos]; r0 = load_prim<PRIMS_LDS>(spheres, quads, e0); }
            while (__ballot(cnt > 0) != 0ull) {
#ifdef MORT_PROFILE_STATES
                gp_liters++; gp_lprims += (unsigned long long)__popcll(__ballot(cnt > 0));
#endif
                if (cnt > 0) {
                    if (cnt > 1) { e1 = entries[pos + 1]; r1 = load_prim<PRIMS_LDS>(spheres, quads, e1); }
                    /* both own roots against the same closest_so_far (independent chains), then applied in entry order: dev_gen.h gen_prim_t */
                    const uint32_t ea = e0, eb = e1;
                    const float ta = gen_prim_t(lsc, chains, rec_sphere(r0), rec_quad(r0), ea, ray, ray_a, closest);
                    float tb = -1.0f;
                    if (cnt > 1) tb = gen_prim_t(lsc, chains, rec_sphere(r1), rec_quad(r1), eb, ray, ray_a, closest);
                    if (cnt > 2) { e0 = entries[pos + 2]; r0 = load_prim<PRIMS_LDS>(spheres, quads, e0); }
                    gen_apply_t(ranks, n_spheres, ta, ea, closest, best, flags);
                    gen_apply_t(ranks, n_spheres, tb, eb, closest, best, flags);
                    pos += 2; cnt -= 2;
                }
            }
            if (state == G_L) {
                if (sp > 0) {
                    sp--;
                    const uint32_t next = tstack[sp * BLOCK];
                    node = next & 0x7fffu;
                    state = (next & 0x8000u) ? G_L : G_T;
                } else { state = (has_media || flags) ? G_M : G_S; kind = K_SHADE; }
            }
            GPROFC(1);
        }
        if (pick == G_M) {
            GPROF(2, nM);
            /* ---- after the solids: the scan itself for undecided rays, then the constant media in scan order
             *      (constant_medium::hit, objects.cuh:396-434; world.cuh:154-160) ---- */
            if (state == G_M) {
                if (flags) {
                    atomicAdd(&gap->f.r.counters[3], 1ull);
                    const ScanHit h = scan_solids(&gap->f.r.sc, first_medium, (const int *)(gap->f.hot_src + gap->o_chains), gap->n_chains, ray.o.x, ray.o.y, ray.o.z,
                                                  ray.d.x, ray.d.y, ray.d.z, ray.tm);
                    best = h.best; closest = h.closest;
                    flags = 0;
                }
                gen_media(lsc, first_medium, n_items, ray, rng, closest, best);
                state = G_S; kind = K_SHADE;
            }
            GPROFC(2);
        }
      }
      if (!running) break;
        {
            GPROF(3, nS);
            /* ---- shade / finish / next sample / next pixel, then start the next ray ---- */
            GPROFS0();
            V3 final_value = mk(0, 0, 0); /* a path's result lives inside one shade step (mega_bvh.h); the degenerate launches enter with K_FINISH and zero */
            if (state == G_S) {
                if (kind == K_SHADE) {
                    if (flags) { /* worlds without media come here directly */
                        atomicAdd(&gap->f.r.counters[3], 1ull);
                        const ScanHit h = scan_solids(&gap->f.r.sc, first_medium, (const int *)(gap->f.hot_src + gap->o_chains), gap->n_chains, ray.o.x, ray.o.y, ray.o.z,
                                                      ray.d.x, ray.d.y, ray.d.z, ray.tm);
                        best = h.best; closest = h.closest;
                        flags = 0;
                    }
#ifdef MORT_DEBUG_PRINT
                    if ((lofs & 0x7fffffff) == L.f.r.debug_lofs) printf("[gen %d seg %u] o (%.9g %.9g %.9g) d (%.9g %.9g %.9g) tm %.9g -> best %08x t %.9g draws %u\n", lofs, segments,
                        ray.o.x, ray.o.y, ray.o.z, ray.d.x, ray.d.y, ray.d.z, ray.tm, best, closest, rng.draws);
#endif
                    if (best == GBEST_NONE) { /* camera.cuh:154-158 */
                        final_value = mk(bg_x, bg_y, bg_z);
                        kind = K_FINISH;
                    } else {
                        const Best b = gen_decode_best(lsc, chains, best, closest);
                        GPROFS(4);
                        /* shade_hit in line: its tables are LDS addresses the compiler can see (ds_read, not flat loads), and nothing is
                         * passed through memory.  (hipcc 7.2's SLP vectorizer miscompiled this form; the library is built without it,
                         * Makefile.)  The parity tests against the oracle are what guards this. */
                        const ShadeOut so = shade_hit(lsc, light_type, light_idx, ray, ray_time0, b, rng);
#ifdef MORT_DEBUG_PRINT
                        if ((lofs & 0x7fffffff) == L.f.r.debug_lofs) printf("   shaded: done %d ident %d o (%.9g %.9g %.9g) d (%.9g %.9g %.9g)\n", (int)so.done, (int)so.ident, ray.o.x, ray.o.y, ray.o.z, ray.d.x, ray.d.y, ray.d.z);
#endif
                        GPROFS(5);
                        if (so.done) { final_value = so.final_value; kind = K_FINISH; }
                        else {
                            if (so.ident) ident_mask |= (1ull << iter);
                            else {
                                float4 e4; e4.x = so.e.kx; e4.y = so.e.ky; e4.z = so.e.kz; e4.w = so.e.rp;
                                if (iter < DL) stack_lds[iter * BLOCK] = e4;
                                else stack_deep[(size_t)(iter - DL) * deep_stride] = e4;
                                /* an entry that is not finite (1 / pdf with pdf = 0) turns even a zero into NaN at the unwind: remember it (bit 31 of lofs) */
                                const float z = e4.x * 0.0f + e4.y * 0.0f + e4.z * 0.0f + e4.w * 0.0f;
                                if (z != z) lofs |= (int)0x80000000u;
                            }
                            iter++;
                            if (iter >= bounce_limit) { final_value = mk(0, 0, 0); kind = K_FINISH; } /* camera.cuh:161-163 */
                        }
                    }
                }
                GPROFS(0);
                if (kind == K_FINISH) { /* unwind + accumulate (camera.cuh:165-173,190); see mega_bvh.h for the identity levels */
                    /* A path that ends in exactly zero radiance -- the bounce limit, a miss of a black background, the back of a light -- unwinds to
                     * exactly (+0, +0, +0) through finite entries: every level is 0 + rp * (k * (+-0)) = +0.  Most paths of the final scene and the
                     * Cornell boxes end that way (the only radiance is the lamp's), and their levels below the LDS part sit in HBM: not read at all. */
                    const bool zero_path = iter > 0 && lofs >= 0 &&
                                           ((__float_as_uint(final_value.x) | __float_as_uint(final_value.y) | __float_as_uint(final_value.z)) & 0x7fffffffu) == 0u;
                    if (zero_path) { final_value = mk(0, 0, 0); iter = 0; }
                    if (iter > 0) {
                        unsigned long long todo = ~ident_mask & (iter >= 64 ? ~0ull : ((1ull << iter) - 1ull));
                        if ((ident_mask >> (iter - 1)) & 1ull) final_value = vadd(mk(0, 0, 0), final_value);
                        /* four levels are fetched before they are applied (deepest first, as the reference unwinds): the deep levels live in
                         * HBM, and one dependent load per level made a long path's unwind a chain of L2 round trips */
                        while (todo != 0ull) {
                            float4 e[4];
                            int n = 0;
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                if (todo != 0ull) {
                                    const int lvl = 63 - __builtin_clzll(todo);
                                    todo &= ~(1ull << lvl);
                                    if (lvl < DL) e[k] = stack_lds[lvl * BLOCK];
                                    else e[k] = stack_deep[(size_t)(lvl - DL) * deep_stride];
                                    n = k + 1;
                                }
                            }
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                if (k < n) {
                                    const V3 t = vmul(mk(e[k].x, e[k].y, e[k].z), final_value);
                                    final_value = vadd(mk(0, 0, 0), vscale(e[k].w, t));
                                }
                            }
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
                        if constexpr (SUB) pixel_write<false, true>(&gap->f, pixel_color.x, pixel_color.y, pixel_color.z, lofs & 0x7fffffff, segments, rng.draws, rng.d, rng.v0, rng.v1, rng.v2, rng.v3, rng.v4);
                        else if (probe) pixel_write<true>(&gap->f, pixel_color.x, pixel_color.y, pixel_color.z, lofs & 0x7fffffff, segments, rng.draws, rng.d, rng.v0, rng.v1, rng.v2, rng.v3, rng.v4);
                        else pixel_write<false>(&gap->f, pixel_color.x, pixel_color.y, pixel_color.z, lofs & 0x7fffffff, segments, rng.draws, rng.d, rng.v0, rng.v1, rng.v2, rng.v3, rng.v4);
                        kind = K_NEWPIX;
                    }
                }
                GPROFS(1);
                if (kind == K_NEWPIX) {
                    const PixelFetch pf = pixel_fetch<SUB>(&gap->f, total_q, role, head_tiles);
                    if (!pf.got) { state = G_DONE; xy = 0; }
                    else {
                        xy = pf.xy; lofs = pf.lofs;
                        rng.d = pf.d; rng.v0 = pf.v0; rng.v1 = pf.v1; rng.v2 = pf.v2; rng.v3 = pf.v3; rng.v4 = pf.v4;
                        rng.draws = 0;
                        pixel_color = mk(0, 0, 0);
                        s_ij = SUB ? ((pf.got - 1) << 16) : 0; segments = 0;
                        kind = (spp > 0) ? K_NEWSAMPLE : K_FINISH;
                        if (spp <= 0) final_value = mk(0, 0, 0);
                    }
                }
                GPROFS(2);
                if (state != G_DONE) {
                    if (kind == K_NEWSAMPLE) { /* camera.cuh:187-190 */
                        lofs &= 0x7fffffff;
                        ray = get_ray(s_cam, xy & 0xffff, (xy >> 16) & 0x7fff, rng, s_ij & 0xffff, s_ij >> 16);
                        ray_time0 = ray.tm;
                        iter = 0;
                        kind = K_SHADE;
                        if (bounce_limit <= 0) { final_value = mk(0, 0, 0); kind = K_FINISH; }
                    }
                    if (kind == K_SHADE) { /* start world::hit for the new ray */
                        ray_a = vlen2(ray.d);
                        const bool ordinary = gen_ray_setup(ray, g_x, g_y, g_z, g_R, g_mnear, g_kmin, gr);
                        closest = __builtin_inff();
                        best = GBEST_NONE;
                        sp = 0;
                        flags = ordinary ? 0 : GFL_REF;
                        segments++;
                        if (!ordinary) state = has_media ? G_M : G_S; /* the scan decides */
                        else if (root == 0xffffu) state = has_media ? G_M : G_S;
                        else { node = root & 0x7fffu; state = (root & 0x8000u) ? G_L : G_T; }
                    }
                }
            }
            /* pool empty (some lane found no pixel) */
            if (!SUB && !draining && __ballot(state == G_DONE) != 0ull) draining = true;
            if (draining && drain_mode == 1 && state != G_DONE) xy |= (int)0x80000000u; /* every pixel still here is a priority pixel */
            GPROFS(3);
            GPROFC(3);
        }
    }
#ifdef MORT_PROFILE_STATES
    if ((threadIdx.x & 63) == 0) {
        unsigned long long *counters = gap->f.r.counters;
        for (int k = 0; k < 4; k++) { atomicAdd(&counters[4 + 2 * k], gp_steps[k]); atomicAdd(&counters[5 + 2 * k], gp_lanes[k]); }
        for (int k = 0; k < 5; k++) atomicAdd(&counters[12 + k], gp_cyc[k]);
        atomicAdd(&counters[17], gp_liters); atomicAdd(&counters[18], gp_lprims);
        for (int k = 0; k < 6; k++) atomicAdd(&counters[20 + k], gp_sp[k]);
        /* MORT_WAVE_LINES=1 in the environment of a profile build: one record per wave -- when it ended (10 ns ticks after its start), its
         * steps and the cycles it spent in each state; the waves that end last are the frame's tail (scripts/wave_lines.py) */
        if (gap->f.wave_log) {
            unsigned long long *w = gap->f.wave_log + 16 * ((size_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6));
            w[0] = blockIdx.x; w[1] = threadIdx.x >> 6; w[2] = __builtin_amdgcn_s_memrealtime() - gp_r0;
            for (int k = 0; k < 4; k++) { w[3 + k] = gp_steps[k]; w[7 + k] = gp_cyc[k]; }
            w[11] = gp_cyc[4];
        }
    }
#endif
}

/* ---- host side ---- */
typedef void (*gen_kernel_t)(const GenArgs);
static gen_kernel_t pick_kernel(int block, bool prims_in_lds, bool sub = false) {
    if (sub) { /* the non-parity launch: 512- and 256-thread workgroups only */
        if (block == 512) return prims_in_lds ? mega_gen_kernel<512, true, true> : mega_gen_kernel<512, false, true>;
        if (block == 256) return prims_in_lds ? mega_gen_kernel<256, true, true> : mega_gen_kernel<256, false, true>;
        return nullptr;
    }
    switch (block) {
    case 1024: return prims_in_lds ? mega_gen_kernel<1024, true> : mega_gen_kernel<1024, false>;
    case 768: return prims_in_lds ? mega_gen_kernel<768, true> : mega_gen_kernel<768, false>;
    case 512: return prims_in_lds ? mega_gen_kernel<512, true> : mega_gen_kernel<512, false>;
    case 256: return prims_in_lds ? mega_gen_kernel<256, true> : mega_gen_kernel<256, false>;
    }
    return nullptr;
}
int mort_gen_blocks_per_cu(int block, bool prims_in_lds, size_t lds_bytes, bool sub) {
    gen_kernel_t k = pick_kernel(block, prims_in_lds, sub);
    if (!k) return 0;
    if (hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return 0;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, block, lds_bytes) != hipSuccess) return 0;
    return per_cu;
}
hipError_t mort_gen_launch(const GenArgs &ga, int block, int grid, size_t lds_bytes, hipStream_t s) {
    gen_kernel_t k = pick_kernel(block, ga.prims_in_lds != 0, ga.f.sub > 0);
    if (!k) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds_bytes, s, ga);
    return hipGetLastError();
}
hipError_t mort_gen_attributes(int block, bool prims_in_lds, hipFuncAttributes *out, bool sub) {
    gen_kernel_t k = pick_kernel(block, prims_in_lds, sub);
    if (!k) return hipErrorInvalidValue;
    return hipFuncGetAttributes(out, (const void *)k);
}
