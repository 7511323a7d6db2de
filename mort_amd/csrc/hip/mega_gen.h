/*
 * mega_gen.h -- launch interface of mega_gen.hip: the state-machine megakernel for worlds WITHOUT reference BVHs
 * (reference scenes 2..9: brute-force world::hit over spheres, quads, instances, lists and constant media,
 * world.cuh:122-168), walking this build's unified tree (scene_compile.h build_unified).
 */
#ifndef MORT_MEGA_GEN_H
#define MORT_MEGA_GEN_H

#include "mega_bvh.h" /* FastArgs, pixel_fetch / pixel_write, thresholds */
#include "dev_gen.h"

struct GenArgs {
    FastArgs f; /* camera / partition / buffers (f.r), work counter, tile order, LDS image source, bounce-stack placement */
    /* offsets of the tables inside the LDS image (f.hot_src, f.hot_bytes) */
    uint32_t o_nodes, o_leaves, o_entries, o_chains, o_xforms, o_items, o_subitems, o_media;
    uint32_t o_lambert, o_metal, o_diel, o_dlight, o_iso, o_solid, o_checker, o_image;
    uint32_t o_spheres, o_quads, o_wspheres, o_wquads, o_ltypes, o_lidxs; /* valid when prims_in_lds */
    uint32_t o_lfirst, o_lcount; /* per hittable_list: first / count in the concatenated list tables (always in the image) */
    int prims_in_lds;
    const uint32_t *ranks; /* scan-order ranks (HBM, behind the LDS part of the image): equal-t ties only */
    int n_spheres;
    uint32_t root;     /* child reference of the tree's root; 0xffff = no solid primitive */
    int first_medium;  /* items[first_medium .. n_items) are constant media */
    int n_chains;      /* entries of the chain table */
    float gx, gy, gz, gR, mnear, kmin; /* far-origin rays widen their error band (scene_compile.h) */
    int th_m;          /* lanes waiting in the media state that trigger a media step */
    int probe;         /* 1 = one-sample cost probe: no image / state output */
    int drain_mode;    /* what a wave does once the pixel pool is empty: 0 = nothing special, 1 = follow its first live lane, 2 = run in rounds (mega_gen.hip) */
};

/* ---- device helpers shared by mega_gen.hip and wave_gen.hip ---- */
/* primitive records that stay in HBM / L2 (worlds whose primitives do not fit in LDS): loaded as global dwordx4, not through a flat
 * pointer (a flat load counts on the LDS counter too, and the leaf step waits on LDS all the time) */
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef const v4f_t __attribute__((address_space(1))) *gv4_ptr;
/* one primitive record as raw registers: a quad's 80 bytes (a..e) or a sphere's 32 (a, b) */
struct PrimRec { v4f_t a, b, c, d, e; };
template <bool IN_LDS> __device__ __forceinline__ PrimRec load_prim(const DSphere *spheres, const DQuad *quads, uint32_t e) {
    PrimRec r;
    const v4f_t z = {0.f, 0.f, 0.f, 0.f};
    r.a = r.b = r.c = r.d = r.e = z;
    if (GENT_QUAD(e)) {
        if (IN_LDS) { const v4f_t *p = (const v4f_t *)(quads + GENT_IDX(e)); r.a = p[0]; r.b = p[1]; r.c = p[2]; r.d = p[3]; r.e = p[4]; }
        else { const gv4_ptr p = (gv4_ptr)(unsigned long long)(quads + GENT_IDX(e)); r.a = p[0]; r.b = p[1]; r.c = p[2]; r.d = p[3]; r.e = p[4]; }
    } else {
        if (IN_LDS) { const v4f_t *p = (const v4f_t *)(spheres + GENT_IDX(e)); r.a = p[0]; r.b = p[1]; }
        else { const gv4_ptr p = (gv4_ptr)(unsigned long long)(spheres + GENT_IDX(e)); r.a = p[0]; r.b = p[1]; }
    }
    return r;
}
__device__ __forceinline__ DSphere rec_sphere(const PrimRec &r) {
    DSphere s;
    s.cx = r.a.x; s.cy = r.a.y; s.cz = r.a.z; s.radius = r.a.w; s.vx = r.b.x; s.vy = r.b.y; s.vz = r.b.z; s.mat = __float_as_uint(r.b.w);
    return s;
}
__device__ __forceinline__ DQuad rec_quad(const PrimRec &r) {
    DQuad q;
    q.Q[0] = r.a.x; q.Q[1] = r.a.y; q.Q[2] = r.a.z; q.D = r.a.w;
    q.u[0] = r.b.x; q.u[1] = r.b.y; q.u[2] = r.b.z; q.area = r.b.w;
    q.v[0] = r.c.x; q.v[1] = r.c.y; q.v[2] = r.c.z; q.mat = __float_as_uint(r.c.w);
    q.n[0] = r.d.x; q.n[1] = r.d.y; q.n[2] = r.d.z; q.pad0 = 0;
    q.w[0] = r.e.x; q.w[1] = r.e.y; q.w[2] = r.e.z; q.pad1 = 0;
    return q;
}

/* host side (mega_gen.hip) */
int mort_gen_blocks_per_cu(int block, bool prims_in_lds, size_t lds_bytes, bool sub = false);
hipError_t mort_gen_launch(const GenArgs &ga, int block, int grid, size_t lds_bytes, hipStream_t s);
hipError_t mort_gen_attributes(int block, bool prims_in_lds, hipFuncAttributes *out, bool sub = false);

#endif
