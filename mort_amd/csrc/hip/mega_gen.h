/*
 * mega_gen.h -- launch interface of mega_gen.hip: the state-machine megakernel for worlds WITHOUT reference BVHs
 * (reference scenes 2..9: brute-force world::hit over spheres, quads, instances, lists and constant media,
 * world.cuh:122-168), walking this build's unified tree (scene_compile.h build_unified).
 */
#ifndef MORT_MEGA_GEN_H
#define MORT_MEGA_GEN_H

#include "mega_bvh.h" /* FastArgs, pixel_fetch / pixel_write, thresholds */

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
};

/* host side (mega_gen.hip) */
int mort_gen_blocks_per_cu(int block, bool prims_in_lds, size_t lds_bytes, bool sub = false);
hipError_t mort_gen_launch(const GenArgs &ga, int block, int grid, size_t lds_bytes, hipStream_t s);
hipError_t mort_gen_attributes(int block, bool prims_in_lds, hipFuncAttributes *out, bool sub = false);

#endif
