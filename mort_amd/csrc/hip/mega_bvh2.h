/*
 * mega_bvh2.h -- hot kernel for BVH scenes (Scene 1, Scene 10), second form.
 *
 * Same arithmetic as mega_bvh.h / the oracle; different scheduling:
 *
 *  - TWO PATHS PER LANE.  Each lane owns two pixels.  Their traversal contexts
 *    (ray, fp64 reciprocals, closest hit, node cursor: 21 VGPRs each) are `c`
 *    (current) and `o` (other); after every step the more runnable one is
 *    swapped into `c` (priority T > L > shade states > done).  A lane whose
 *    current path is waiting for its shade step keeps traversing its other
 *    path, so box steps run with most lanes active while shade work piles up
 *    into large batches.
 *  - SHADE SPLIT BY MATERIAL CLASS.  Separate steps for diffuse (lambertian /
 *    isotropic: cosine sampling, two fp64 sincos), specular / emissive (metal,
 *    dielectric, diffuse_light) and finish (miss or terminated path: unwind,
 *    accumulate, next sample or next pixel, camera ray).  A step runs only the
 *    class with the most waiting lanes, so a batch does not pay for every
 *    branch of the reference's material switch (materials.cuh:272-296).
 *  - BOUNCE STACK IN LDS.  16 B entries (scattering_pdf * attenuation, 1/pdf)
 *    at [path][depth][thread]: conflict-free ds_write/read_b128; depths beyond
 *    the LDS budget spill to a per-path HBM array (rare: deep glass paths).
 *    The reference keeps bounce_limit * W * H * 32 B in HBM (mort.cu:712-725).
 *  - Scene hot blob in LDS, pixels fetched from an atomic counter in 8x8-tile
 *    order, aabb::hit with per-ray reciprocals and v_max/min updates: as in
 *    mega_bvh.h.
 */
#ifndef MORT_MEGA_BVH2_H
#define MORT_MEGA_BVH2_H

#include "dev_trace.h"

struct Fast2Args {
    RenderArgs r;
    const unsigned char *hot_src;
    uint32_t hot_bytes;
    uint32_t off_nodes, off_spheres, off_lambert, off_metal, off_diel, off_dlight, off_iso, off_solid, off_checker;
    uint32_t off_stack;    /* LDS offset of the bounce stacks */
    int stack_lds_depth;   /* entries per path kept in LDS */
    float4 *stack_ovf;     /* HBM overflow: [(block * BLOCK + tid) * 2 + pid][MORT_MAX_BOUNCE_LIMIT] */
    int node_first, node_count;
    unsigned int *next_q;
    int tiles_x, tiles_total;
};

enum { P_DONE = 0, P_FIN = 1, P_SPEC = 2, P_LAMB = 3, P_L = 4, P_T = 5 };

struct Trav {
    float ox, oy, oz, dx, dy, dz, tm; /* ray; once a path has terminated (P_FIN) dx,dy,dz carry its final value */
    double ix, iy, iz;                /* 1.0 / dir, fp64 (aabb.cuh:40) */
    float jx, jy, jz;                 /* fl32 of the reciprocals (slab_hit) */
    int exact_only;
    float a, closest;
    int best;
    uint32_t bmat;
    int node;
    uint32_t leaf;
    int state;
    int pid;
};

struct Pix {
    uint32_t d, v0, v1, v2, v3, v4, draws;
    float cr, cg, cb;
    int s_i, s_j, iter, lofs, x, y;
    uint32_t segments;
    float time0;
};

#define SWAPF(A, B) do { float t_ = (A); (A) = (B); (B) = t_; } while (0)
#define SWAPI(A, B) do { int t_ = (A); (A) = (B); (B) = t_; } while (0)
#define SWAPU(A, B) do { uint32_t t_ = (A); (A) = (B); (B) = t_; } while (0)
#define SWAPD(A, B) do { double t_ = (A); (A) = (B); (B) = t_; } while (0)

DEV void swap_trav(Trav &a, Trav &b) {
    SWAPF(a.ox, b.ox); SWAPF(a.oy, b.oy); SWAPF(a.oz, b.oz); SWAPF(a.dx, b.dx); SWAPF(a.dy, b.dy); SWAPF(a.dz, b.dz); SWAPF(a.tm, b.tm);
    SWAPD(a.ix, b.ix); SWAPD(a.iy, b.iy); SWAPD(a.iz, b.iz); SWAPF(a.jx, b.jx); SWAPF(a.jy, b.jy); SWAPF(a.jz, b.jz); SWAPI(a.exact_only, b.exact_only);
    SWAPF(a.a, b.a); SWAPF(a.closest, b.closest);
    SWAPI(a.best, b.best); SWAPU(a.bmat, b.bmat); SWAPI(a.node, b.node); SWAPU(a.leaf, b.leaf); SWAPI(a.state, b.state); SWAPI(a.pid, b.pid);
}

#define PIX_FIELDS(X) X(d) X(v0) X(v1) X(v2) X(v3) X(v4) X(draws) X(cr) X(cg) X(cb) X(s_i) X(s_j) X(iter) X(lofs) X(x) X(y) X(segments) X(time0)
DEV Pix pix_load(const Pix &p0, const Pix &p1, int pid) {
    Pix r;
#define X(f) r.f = pid ? p1.f : p0.f;
    PIX_FIELDS(X)
#undef X
    return r;
}
DEV void pix_store(Pix &p0, Pix &p1, int pid, const Pix &r) {
#define X(f) p0.f = pid ? p0.f : r.f; p1.f = pid ? r.f : p1.f;
    PIX_FIELDS(X)
#undef X
}

#ifndef MORT2_TH_T
#define MORT2_TH_T 40
#endif
#ifndef MORT2_BLOCK
#define MORT2_BLOCK 512
#endif
#ifndef MORT2_MIN_WAVES
#define MORT2_MIN_WAVES 2
#endif

template <int BLOCK, int TH_T>
__global__ void __launch_bounds__(BLOCK, MORT2_MIN_WAVES) mega_bvh2_kernel(const Fast2Args fa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const RenderArgs &a = fa.r;
    {
        const uint4 *src = (const uint4 *)fa.hot_src;
        uint4 *dst = (uint4 *)lds;
        const uint32_t n16 = fa.hot_bytes >> 4;
        for (uint32_t i = threadIdx.x; i < n16; i += BLOCK) dst[i] = src[i];
    }
    __syncthreads();
    const DBvhNode *nodes = (const DBvhNode *)(lds + fa.off_nodes);
    const DSphere *spheres = (const DSphere *)(lds + fa.off_spheres);
    const DLambert *lambert = (const DLambert *)(lds + fa.off_lambert);
    const DMetal *metal = (const DMetal *)(lds + fa.off_metal);
    const DDielectric *dielectric = (const DDielectric *)(lds + fa.off_diel);
    const DLambert *dlight = (const DLambert *)(lds + fa.off_dlight);
    const DLambert *isotropic = (const DLambert *)(lds + fa.off_iso);
    const DSolid *solid = (const DSolid *)(lds + fa.off_solid);
    const DChecker *checker = (const DChecker *)(lds + fa.off_checker);
    float4 *stack_lds = (float4 *)(lds + fa.off_stack);
    const int DL = fa.stack_lds_depth;
    float4 *stack_ovf = fa.stack_ovf + ((size_t)blockIdx.x * BLOCK + threadIdx.x) * 2 * MORT_MAX_BOUNCE_LIMIT;

    const int node_first = fa.node_first, node_end = fa.node_first + fa.node_count;
    const int sq = a.sqrt_spp;
    const unsigned total_q = (unsigned)fa.tiles_total * 64u;
    const int tid = threadIdx.x;

    Trav c, o;
    c.ox = c.oy = c.oz = 0; c.dx = c.dy = 0; c.dz = 1; c.tm = 0; c.ix = c.iy = c.iz = 1; c.jx = c.jy = c.jz = 1; c.exact_only = 0; c.a = 1; c.closest = 0;
    c.best = -1; c.bmat = 0; c.node = 0; c.leaf = 0; c.state = P_FIN; c.pid = 0;
    o = c; o.pid = 1;
    Pix p0, p1;
    p0.d = p0.v0 = p0.v1 = p0.v2 = p0.v3 = p0.v4 = p0.draws = 0; p0.cr = p0.cg = p0.cb = 0;
    p0.s_i = 0; p0.s_j = -1; /* s_j < 0: no pixel yet */
    p0.iter = 0; p0.lofs = 0; p0.x = p0.y = 0; p0.segments = 0; p0.time0 = 0;
    p1 = p0;
    unsigned long long tot_segments = 0, tot_draws = 0;
#ifdef MORT_PROFILE_STATES
    unsigned long long prof_steps[6] = {0, 0, 0, 0, 0, 0}, prof_lanes[6] = {0, 0, 0, 0, 0, 0}, prof_cyc[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long pt0 = __builtin_readcyclecounter(), pt1;
#define PROF2(i, lanes) do { prof_steps[i] += 1; prof_lanes[i] += (unsigned long long)(lanes); } while (0)
#define PROFC2(i) do { pt1 = __builtin_readcyclecounter(); prof_cyc[i] += pt1 - pt0; pt0 = pt1; } while (0)
#else
#define PROF2(i, lanes) do { } while (0)
#define PROFC2(i) do { } while (0)
#endif

#define STACK_PUT(pid_, depth_, val_) do { \
        if ((depth_) < DL) stack_lds[((pid_) * DL + (depth_)) * BLOCK + tid] = (val_); \
        else stack_ovf[(pid_) * MORT_MAX_BOUNCE_LIMIT + (depth_)] = (val_); } while (0)
#define STACK_GET(pid_, depth_) (((depth_) < DL) ? stack_lds[((pid_) * DL + (depth_)) * BLOCK + tid] : stack_ovf[(pid_) * MORT_MAX_BOUNCE_LIMIT + (depth_)])

    /* start world::hit for the ray in c (world.cuh:104-120: one BVH, bvh_mode) */
#define BEGIN_TRAVERSAL(P) do { \
        { const SlabRay sr_ = slab_ray(c.ox, c.oy, c.oz, c.dx, c.dy, c.dz); c.ix = sr_.ix; c.iy = sr_.iy; c.iz = sr_.iz; \
          c.jx = sr_.i32x; c.jy = sr_.i32y; c.jz = sr_.i32z; c.exact_only = sr_.exact_only ? 1 : 0; } \
        c.a = c.dx * c.dx + c.dy * c.dy + c.dz * c.dz; \
        c.closest = __builtin_inff(); c.best = -1; c.bmat = 0; c.node = node_first; \
        (P).segments++; \
        c.state = (node_first < node_end) ? P_T : P_FIN; \
        if (node_first >= node_end) { c.dx = a.background.x; c.dy = a.background.y; c.dz = a.background.z; } } while (0)

    /* traversal finished: classify the shade step by the winner's material */
#define END_TRAVERSAL() do { \
        if (c.best < 0) { c.state = P_FIN; c.dx = a.background.x; c.dy = a.background.y; c.dz = a.background.z; } \
        else { const int mt_ = DREF_TYPE(c.bmat); c.state = (mt_ == MORT_MAT_LAMBERTIAN || mt_ == MORT_MAT_ISOTROPIC) ? P_LAMB : P_SPEC; } } while (0)

    for (;;) {
        if (o.state > c.state) swap_trav(c, o);
        const int nT = __popcll(__ballot(c.state == P_T));
        int pick;
        if (nT >= TH_T) {
            pick = P_T;
        } else {
            if (__ballot(c.state != P_DONE) == 0ull) break;
            const int nL = __popcll(__ballot(c.state == P_L));
            const int nLam = __popcll(__ballot(c.state == P_LAMB || o.state == P_LAMB));
            const int nSpec = __popcll(__ballot(c.state == P_SPEC || o.state == P_SPEC));
            const int nFin = __popcll(__ballot(c.state == P_FIN || o.state == P_FIN));
            int bestn = nL; pick = P_L;
            if (nLam > bestn) { bestn = nLam; pick = P_LAMB; }
            if (nFin > bestn) { bestn = nFin; pick = P_FIN; }
            if (nSpec > bestn) { bestn = nSpec; pick = P_SPEC; }
            if (bestn == 0) pick = P_T;
        }
        PROFC2(0);

        if (pick == P_T) {
            int keep;
            do {
                PROF2(P_T, __popcll(__ballot(c.state == P_T)));
                if (c.state == P_T) {
                    const DBvhNode nd = nodes[c.node];
                    SlabRay sr; sr.ox = c.ox; sr.oy = c.oy; sr.oz = c.oz; sr.ix = c.ix; sr.iy = c.iy; sr.iz = c.iz;
                    sr.i32x = c.jx; sr.i32y = c.jy; sr.i32z = c.jz; sr.exact_only = c.exact_only != 0;
                    const bool miss = !slab_hit(nd, sr, c.closest);
                    const int skip = (int)(nd.skip & 0x7fffffffu);
                    if (miss) {
                        c.node = skip;
                    } else if (nd.skip >> 31) {
                        c.leaf = nd.prims;
                        c.node = skip;
                        c.state = P_L;
                    } else {
                        c.node = c.node + 1;
                    }
                    if (c.state == P_T && c.node >= node_end) END_TRAVERSAL();
                }
                keep = __popcll(__ballot(c.state == P_T));
            } while (keep >= TH_T);
            PROFC2(P_T);
        } else if (pick == P_L) {
            PROF2(P_L, __popcll(__ballot(c.state == P_L)));
            if (c.state == P_L) { /* sphere::hit on the leaf's one or two spheres (objects.cuh:60-77,690-692) */
                const uint32_t pa = c.leaf & 0x7fffu, pb = (c.leaf >> 16) & 0x7fffu;
                Ray r; r.o = mk(c.ox, c.oy, c.oz); r.d = mk(c.dx, c.dy, c.dz); r.tm = c.tm;
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const uint32_t p = k ? pb : pa;
                    if (k == 1 && pb == pa) break;
                    const DSphere sp = spheres[p];
                    float t;
                    if (sphere_hit_t(sp, r, c.a, 0.001f, c.closest, t)) { c.closest = t; c.best = (int)p; c.bmat = sp.mat; }
                }
                if (c.node >= node_end) END_TRAVERSAL(); else c.state = P_T;
            }
            PROFC2(P_L);
        } else if (pick == P_LAMB) {
            PROF2(P_LAMB, __popcll(__ballot(c.state == P_LAMB || o.state == P_LAMB)));
            /* ---- diffuse: lambertian / isotropic scatter + cosine / sphere pdf (materials.cuh:38-55,182-198; pdf.cuh:29-54) ---- */
            if (c.state == P_LAMB || o.state == P_LAMB) {
                if (c.state != P_LAMB) swap_trav(c, o);
                Pix P = pix_load(p0, p1, c.pid);
                Rng rng; rng.d = P.d; rng.v0 = P.v0; rng.v1 = P.v1; rng.v2 = P.v2; rng.v3 = P.v3; rng.v4 = P.v4; rng.draws = P.draws;
                Ray ray; ray.o = mk(c.ox, c.oy, c.oz); ray.d = mk(c.dx, c.dy, c.dz); ray.tm = c.tm;
                const DSphere sp = spheres[c.best];
                const V3 p = ray_at(ray, c.closest);
                const V3 outward = vdiv(vsub(p, sphere_center(sp, ray.tm)), sp.radius);
                const bool front_face = vdot(ray.d, outward) < 0;
                const V3 normal = front_face ? outward : vneg(outward);
                const int mtype = DREF_TYPE(sp.mat), midx = DREF_IDX(sp.mat);
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
                            const DChecker ck = checker[ti];
                            const int xi = mort_f2i(mort_floorf(ck.inv_scale * p.x));
                            const int yi = mort_f2i(mort_floorf(ck.inv_scale * p.y));
                            const int zi = mort_f2i(mort_floorf(ck.inv_scale * p.z));
                            tex = ((xi + yi + zi) % 2 == 0) ? ck.even : ck.odd;
                        } else break;
                    }
                    if (!resolved) { float u, v; sphere_uv(outward, u, v); attenuation = texture_value(a.sc, tex, u, v, p); }
                }
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
                float4 e;
                e.x = scattering_pdf * attenuation.x; e.y = scattering_pdf * attenuation.y; e.z = scattering_pdf * attenuation.z;
                e.w = 1 / mat_pdf;
                STACK_PUT(c.pid, P.iter, e);
                P.iter++;
                c.ox = p.x; c.oy = p.y; c.oz = p.z; c.dx = dir.x; c.dy = dir.y; c.dz = dir.z; c.tm = P.time0;
                if (P.iter >= a.bounce_limit) { c.state = P_FIN; c.dx = 0; c.dy = 0; c.dz = 0; } /* camera.cuh:161-163 */
                else BEGIN_TRAVERSAL(P);
                P.d = rng.d; P.v0 = rng.v0; P.v1 = rng.v1; P.v2 = rng.v2; P.v3 = rng.v3; P.v4 = rng.v4; P.draws = rng.draws;
                pix_store(p0, p1, c.pid, P);
            }
            PROFC2(P_LAMB);
        } else if (pick == P_SPEC) {
            PROF2(P_SPEC, __popcll(__ballot(c.state == P_SPEC || o.state == P_SPEC)));
            /* ---- metal / dielectric (skip_pdf rays), diffuse_light and unknown tags (materials.cuh:73-163) ---- */
            if (c.state == P_SPEC || o.state == P_SPEC) {
                if (c.state != P_SPEC) swap_trav(c, o);
                Pix P = pix_load(p0, p1, c.pid);
                Rng rng; rng.d = P.d; rng.v0 = P.v0; rng.v1 = P.v1; rng.v2 = P.v2; rng.v3 = P.v3; rng.v4 = P.v4; rng.draws = P.draws;
                Ray ray; ray.o = mk(c.ox, c.oy, c.oz); ray.d = mk(c.dx, c.dy, c.dz); ray.tm = c.tm;
                const DSphere sp = spheres[c.best];
                const V3 p = ray_at(ray, c.closest);
                const V3 outward = vdiv(vsub(p, sphere_center(sp, ray.tm)), sp.radius);
                const bool front_face = vdot(ray.d, outward) < 0;
                const V3 normal = front_face ? outward : vneg(outward);
                const int mtype = DREF_TYPE(sp.mat), midx = DREF_IDX(sp.mat);
                if (mtype == MORT_MAT_METAL || mtype == MORT_MAT_DIELECTRIC) {
                    V3 ndir;
                    float4 e;
                    if (mtype == MORT_MAT_METAL) {
                        const DMetal m = metal[midx];
                        V3 reflected = reflect(ray.d, normal);
                        ndir = vadd(vunit(reflected), vscale(m.fuzz, random_unit_vector(rng)));
                        e.x = 1.0f * m.r; e.y = 1.0f * m.g; e.z = 1.0f * m.b; e.w = 1.0f;
                    } else {
                        const DDielectric m = dielectric[midx];
                        const float refraction_ratio = front_face ? m.inv_ior : m.ior;
                        const V3 unit_direction = vunit(ray.d);
                        const float cos_theta = (float)mort_fmin((double)vdot(vneg(unit_direction), normal), 1.0);
                        const float sin_theta = (float)mort_sqrt(1.0 - (double)(cos_theta * cos_theta));
                        const bool cant_refract = (double)(refraction_ratio * sin_theta) > 1.0;
                        if (cant_refract || reflectance(cos_theta, refraction_ratio) > random_float(rng))
                            ndir = reflect(unit_direction, normal);
                        else
                            ndir = refract(unit_direction, normal, refraction_ratio);
                        e.x = 1.0f; e.y = 1.0f; e.z = 1.0f; e.w = 1.0f;
                    }
                    STACK_PUT(c.pid, P.iter, e);
                    P.iter++;
                    c.ox = p.x; c.oy = p.y; c.oz = p.z; c.dx = ndir.x; c.dy = ndir.y; c.dz = ndir.z; /* time stays r_in.time() */
                    if (P.iter >= a.bounce_limit) { c.state = P_FIN; c.dx = 0; c.dy = 0; c.dz = 0; }
                    else BEGIN_TRAVERSAL(P);
                } else {
                    V3 emission = mk(0, 0, 0);
                    if (mtype == MORT_MAT_DIFFUSE_LIGHT && front_face) {
                        const DLambert m = dlight[midx];
                        if (m.tex == 0) emission = mk(m.r, m.g, m.b);
                        else { float u, v; sphere_uv(outward, u, v); emission = texture_value(a.sc, m.tex, u, v, p); }
                    }
                    c.state = P_FIN; c.dx = emission.x; c.dy = emission.y; c.dz = emission.z;
                }
                P.d = rng.d; P.v0 = rng.v0; P.v1 = rng.v1; P.v2 = rng.v2; P.v3 = rng.v3; P.v4 = rng.v4; P.draws = rng.draws;
                pix_store(p0, p1, c.pid, P);
            }
            PROFC2(P_SPEC);
        } else {
            PROF2(P_FIN, __popcll(__ballot(c.state == P_FIN || o.state == P_FIN)));
            /* ---- finish: unwind + accumulate (camera.cuh:165-173,190), next sample or pixel, camera ray ---- */
            if (c.state == P_FIN || o.state == P_FIN) {
                if (c.state != P_FIN) swap_trav(c, o);
                Pix P = pix_load(p0, p1, c.pid);
                bool need_pixel = (P.s_j < 0);
                if (!need_pixel) {
                    V3 fv = mk(c.dx, c.dy, c.dz);
                    while (P.iter > 0) {
                        P.iter--;
                        const float4 e = STACK_GET(c.pid, P.iter);
                        const V3 t = vmul(mk(e.x, e.y, e.z), fv);
                        fv = vadd(mk(0, 0, 0), vscale(e.w, t));
                    }
                    P.cr += fv.x; P.cg += fv.y; P.cb += fv.z;
                    P.s_i++;
                    if (P.s_i >= sq) { P.s_i = 0; P.s_j++; }
                    if (P.s_j >= sq) { /* pixel complete: camera.cuh:194-207 */
                        V3 col = vscale(a.pixel_samples_scale, mk(P.cr, P.cg, P.cb));
                        if (col.x != col.x) col.x = 0.0f;
                        if (col.y != col.y) col.y = 0.0f;
                        if (col.z != col.z) col.z = 0.0f;
                        if (a.accum) { a.accum[3 * P.lofs] = col.x; a.accum[3 * P.lofs + 1] = col.y; a.accum[3 * P.lofs + 2] = col.z; }
                        float g[3] = {mort_sqrtf(col.x), mort_sqrtf(col.y), mort_sqrtf(col.z)};
                        unsigned char b[3];
#pragma unroll
                        for (int k = 0; k < 3; k++) {
                            float v = g[k];
                            if (v < 0.0f) v = 0.0f;
                            if (v > 0.999f) v = 0.999f;
                            b[k] = (unsigned char)mort_f2i(256 * v);
                        }
                        uchar4 out; out.x = b[0]; out.y = b[1]; out.z = b[2]; out.w = 255;
                        a.rgba[P.lofs] = out;
                        if (a.seg_px) a.seg_px[P.lofs] = P.segments;
                        mort_rng_state st;
                        st.d = P.d; st.v[0] = P.v0; st.v[1] = P.v1; st.v[2] = P.v2; st.v[3] = P.v3; st.v[4] = P.v4;
                        st.boxmuller_flag = 0; st.boxmuller_flag_double = 0; st.boxmuller_extra = 0.f; st.boxmuller_extra_double = 0.;
                        a.states[P.lofs] = st;
                        tot_segments += P.segments; tot_draws += P.draws;
                        need_pixel = true;
                    }
                }
                bool alive = true;
                if (need_pixel) {
                    bool got = false;
                    while (!got) {
                        const unsigned long long need = __ballot(1);
                        const int cnt = __popcll(need);
                        const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(need >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need, 0));
                        unsigned base = 0;
                        if (rank == 0) base = atomicAdd(fa.next_q, (unsigned)cnt);
                        base = __shfl(base, __ffsll((long long)need) - 1);
                        const unsigned q = base + (unsigned)rank;
                        if (q >= total_q) { alive = false; break; }
                        const int tile = (int)(q >> 6), within = (int)(q & 63u);
                        const int tx = tile % fa.tiles_x, ty = tile / fa.tiles_x;
                        const int qx = tx * 8 + (within & 7), qly = ty * 8 + (within >> 3);
                        if (qx < a.width && qly < a.local_rows) {
                            P.x = qx;
                            P.y = global_row(qly, a.rank, a.nranks, a.rows_per_block);
                            P.lofs = qx + qly * a.width;
                            got = true;
                        }
                    }
                    if (alive) {
                        const mort_rng_state st = a.states[P.lofs];
                        P.d = st.d; P.v0 = st.v[0]; P.v1 = st.v[1]; P.v2 = st.v[2]; P.v3 = st.v[3]; P.v4 = st.v[4];
                        P.draws = 0; P.cr = P.cg = P.cb = 0; P.s_i = 0; P.s_j = 0; P.segments = 0; P.iter = 0;
                    } else {
                        P.s_j = -1;
                        c.state = P_DONE;
                    }
                }
                if (alive) { /* camera.cuh:187-190 (sqrt_spp >= 1 and bounce_limit >= 1 on this path: checked on the host) */
                    Rng rng; rng.d = P.d; rng.v0 = P.v0; rng.v1 = P.v1; rng.v2 = P.v2; rng.v3 = P.v3; rng.v4 = P.v4; rng.draws = P.draws;
                    const Ray ray = get_ray(a, P.x, P.y, rng, P.s_i, P.s_j);
                    P.d = rng.d; P.v0 = rng.v0; P.v1 = rng.v1; P.v2 = rng.v2; P.v3 = rng.v3; P.v4 = rng.v4; P.draws = rng.draws;
                    P.time0 = ray.tm;
                    P.iter = 0;
                    c.ox = ray.o.x; c.oy = ray.o.y; c.oz = ray.o.z; c.dx = ray.d.x; c.dy = ray.d.y; c.dz = ray.d.z; c.tm = ray.tm;
                    BEGIN_TRAVERSAL(P);
                }
                pix_store(p0, p1, c.pid, P);
            }
            PROFC2(P_FIN);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        tot_segments += __shfl_down(tot_segments, off);
        tot_draws += __shfl_down(tot_draws, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&a.counters[0], tot_segments);
        atomicAdd(&a.counters[1], tot_draws);
#ifdef MORT_PROFILE_STATES
        for (int k = 0; k < 6; k++) { atomicAdd(&a.counters[4 + k], prof_steps[k]); atomicAdd(&a.counters[10 + k], prof_lanes[k]); atomicAdd(&a.counters[16 + k], prof_cyc[k]); }
#endif
    }
#undef STACK_PUT
#undef STACK_GET
#undef BEGIN_TRAVERSAL
#undef END_TRAVERSAL
}

#endif
