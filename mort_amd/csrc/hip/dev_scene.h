/*
 * dev_scene.h -- the scene as the gfx950 kernels see it.
 *
 * mort_hip_upload_world() "compiles" the reference's tagged object graph
 * (world.cuh:104-171 walks it with switch dispatch and recursion) into flat,
 * 16-byte-aligned arrays sized for LDS staging:
 *
 *   - the closest-hit sequence of world::hit becomes a list of ITEMS in the
 *     reference's visiting order (BVHs, then spheres, quads, translates,
 *     rotate_ys, constant_mediums, lists; world.cuh:110-168); nested
 *     translate / rotate_y / hittable_list objects are flattened into runs of
 *     primitives that share a transform CHAIN;
 *   - each BVH (objects.cuh:725-735, SoA with child type/index arrays) becomes
 *     a threaded pre-order array: a box hit moves to node+1, a miss (or a
 *     finished leaf) to node.skip, which reproduces the reference's left-first
 *     stack walk (objects.cuh:664-723) without a stack;
 *   - primitives keep only the fields hit/pdf/random read.
 */
#ifndef MORT_DEV_SCENE_H
#define MORT_DEV_SCENE_H

#include <stdint.h>
#include "mort_scene.h"

/* material / texture references: (type << 16) | idx */
#define DREF(type, idx) (((uint32_t)(type) << 16) | (uint32_t)(idx))
#define DREF_TYPE(r) ((int)((r) >> 16) & 0x7fff)
#define DREF_IDX(r) ((int)((r) & 0xffffu))

struct __attribute__((aligned(16))) DSphere { /* 32 B */
    float cx, cy, cz, radius;
    float vx, vy, vz;   /* center_vec (0 when !moves) */
    uint32_t mat;       /* DREF | moves << 31 */
};

struct __attribute__((aligned(16))) DQuad { /* 80 B */
    float Q[3], D;
    float u[3], area;
    float v[3]; uint32_t mat;
    float n[3]; uint32_t pad0;
    float w[3]; uint32_t pad1;
};

/* threaded BVH node, pre-order; 32 B */
struct __attribute__((aligned(16))) DBvhNode {
    float xmin, xmax, ymin, ymax, zmin, zmax;
    uint32_t skip;  /* next node when this subtree is done/missed; bit 31 = leaf */
    uint32_t prims; /* leaf: primA | primB << 16; prim = kind << 15 | index (kind 0 sphere, 1 quad) */
};

/* Node of this build's OWN tree over the reference's leaf nodes (mega_bvh.h; scene_compile.h build_own_tree): the
 * boxes of its two children, 64 B = four ds_read_b128.  A child reference is a node index, or 0x8000 | leaf index
 * (leaf records are DBvhNode: the reference leaf node's box, bit for bit, and its one or two spheres).
 * e0/e1: anomaly margin of the child in world units (largest leaf-box diagonal below it + 8e-3 x its own diagonal). */
struct __attribute__((aligned(16))) DNode2 {
    float x0min, x0max, y0min, y0max;
    float z0min, z0max, x1min, x1max;
    float y1min, y1max, z1min, z1max;
    uint32_t child0, child1;
    float e0, e1;
};
/* The same tree four children wide (scene_compile.h collapse_own_tree; the BVH megakernel's box step): 128 B = eight ds_read_b128,
 * child k's planes at [k] of each array.  A binary node's larger inner child is opened in place until four slots are full, so
 * one step tests what two to three binary steps did and the loads of a step are all independent.  Child references are
 * pre-scaled (below); an unused slot is 0xffff (never entered).  Boxes and margins are the binary tree's, bit for bit. */
struct __attribute__((aligned(16))) DNode4 {
    float xmin[4], xmax[4], ymin[4], ymax[4], zmin[4], zmax[4], e[4];
    uint32_t child[4];
    uint32_t pad[4]; /* 144 B = 9 x 16: the lanes of a wave read the same 16-byte piece of DIFFERENT nodes, and a ds_read_b128 serves 16 lanes per LDS cycle from
                      * 64 banks.  With a stride of 128 B those pieces lie on two bank positions (8-way conflicts, measured: 45 % of the kernel's LDS cycles);
                      * an odd number of pieces per node spreads them over all sixteen */
};
/* What the leaf step of the BVH megakernel reads, in ONE round trip to LDS: the one or two spheres of a reference leaf node by value;
 * then the leaf node's own box, bit for bit, for the final check of the winner.  112 B = 7 x 16, an odd number of pieces again.
 * References to both kinds of record are pre-scaled to 16-byte pieces so that an address is one shift-add: an inner child is
 * MORT_NODE4_PIECES x its node index, a leaf child 0x8000 | MORT_LEAF2_PIECES x its leaf index. */
struct __attribute__((aligned(16))) DLeaf2 {
    DSphere a, b;     /* b == a when the leaf holds one sphere */
    float xmin, xmax, ymin, ymax, zmin, zmax;
    uint32_t n;       /* 1 or 2 */
    uint32_t prims;   /* pa | pb << 16: their indices in the scene's sphere table */
    uint32_t pad[4];
};
#define MORT_NODE4_PIECES 9
#define MORT_LEAF2_PIECES 7
#define MORT_OWN4_STACK 24 /* pending children per lane of the four-wide walk: a step pushes at most 3; the builder checks the worst path */
/* Node of the UNIFIED tree (scene_compile.h build_unified): 32 B = two ds_read_b128.  The boxes of its two children as 8-bit offsets
 * from the node's own corner, in steps of a power of two per axis: plane = fmaf((float)q, step, origin).  The builder rounds every
 * plane OUTWARD and checks it with the same fmaf, so a decoded box contains the child's padded box and every argument of
 * build_unified's header holds for it (the walk only ever needs boxes that are not too small).  Half the bytes of DNode2: the final
 * scene's 3 408 primitives get leaves of at most 2 primitives in fewer LDS bytes than leaves of 4 took before.
 * A child reference is 16 bits: an inner node index, or 0x8000 | (primitives - 1) << 13 | first entry (a leaf is a run of entries). */
struct __attribute__((aligned(16))) DNodeQ {
    float ox, oy, oz;
    uint32_t exps;     /* biased exponents of the steps: x | y << 8 | z << 16 */
    uint32_t q0;       /* child 0: xlo | xhi << 8 | ylo << 16 | yhi << 24 */
    uint32_t q1;       /* child 0: zlo | zhi << 8 | child 1: xlo << 16 | xhi << 24 */
    uint32_t q2;       /* child 1: ylo | yhi << 8 | zlo << 16 | zhi << 24 */
    uint32_t children; /* child 0 | child 1 << 16 */
};
#define GLEAF_FIRST(ref) ((ref) & 0x1fffu)
#define GLEAF_COUNT(ref) ((((ref) >> 13) & 3u) + 1u)
#define MORT_OWN_STACK 16 /* pending far children per lane kept in LDS; deeper walks use the reference walk */
#define MORT_OWN_MAX_DEPTH 15
/* unified tree (scene_compile.h build_unified): at most this many primitives per leaf; an entry names one primitive */
#ifndef MORT_GEN_LEAF_MAX
#define MORT_GEN_LEAF_MAX 2 /* at most 4 (two bits of a leaf reference).  Measured on the final scene with 64-byte nodes (leaves of 3 / 4 / 6): 320 / 353 / 394 ms at
                              * 800x800x100, 189 / 198 / 214 ms at 1920x1080x49; leaves of 2 need the 32-byte nodes to fit LDS */
#endif
#define GENT(kind, chain, idx) (((uint32_t)(kind) << 31) | ((uint32_t)(chain) << 24) | (uint32_t)(idx))
#define GENT_QUAD(e) ((e) >> 31)
#define GENT_CHAIN(e) (((e) >> 24) & 0x7fu)
#define GENT_IDX(e) ((e) & 0xffffffu)

enum { XF_TRANSLATE = 0, XF_ROTATE_Y = 1 };
struct __attribute__((aligned(16))) DXform { /* 16 B */
    int kind;
    float a, b, c; /* translate: offset xyz; rotate_y: sin, cos, - */
};

enum { ITEM_BVH = 0, ITEM_SPHERES = 1, ITEM_QUADS = 2, ITEM_MEDIUM = 3 };
struct __attribute__((aligned(16))) DItem { /* 32 B */
    int kind;
    int first, count;          /* BVH: node range; prims: range in dev arrays; medium: sub-item range */
    int chain_first, chain_count;
    uint32_t mat;              /* medium: phase-function material */
    int medium;                /* medium: index into neg_inv_density[] */
    int pad0;
};

struct __attribute__((aligned(16))) DLambert { float r, g, b; uint32_t tex; }; /* tex != 0: look up, else rgb inline */
struct __attribute__((aligned(16))) DMetal { float r, g, b, fuzz; };
struct __attribute__((aligned(8))) DDielectric { float ior, inv_ior; };
struct __attribute__((aligned(16))) DSolid { float r, g, b, pad; };
struct __attribute__((aligned(16))) DChecker { float inv_scale; uint32_t even, odd; uint32_t pad; };
struct __attribute__((aligned(16))) DImage { uint32_t offset; int width, height, pad; }; /* offset into texels */

/* light object for hittable_pdf (pdf.cuh:60-80): a flattened list of
 * sphere / quad primitives, or "none of those" (value 0, direction (1,0,0)). */
enum { LIGHT_NONE = -1, LIGHT_INVALID = 0, LIGHT_SPHERE = 1, LIGHT_QUAD = 2, LIGHT_LIST = 3 };

/* All pointers are device addresses inside one allocation. */
struct DScene {
    const DItem *items;     int n_items;      /* world-level sequence */
    const DItem *subitems;  int n_subitems;   /* medium boundaries */
    const DBvhNode *nodes;  int n_nodes;
    const DSphere *spheres; int n_spheres;
    const DQuad *quads;     int n_quads;
    const DXform *xforms;   int n_xforms;
    const double *neg_inv_density; int n_media;
    const DLambert *lambert; const DMetal *metal; const DDielectric *dielectric;
    const DLambert *dlight;  const DLambert *isotropic; /* {rgb | tex} like lambertian */
    const DSolid *solid; const DChecker *checker; const DImage *image;
    const DImage *image_hbm; /* == image; stays the HBM copy in a kernel's LDS view of the scene (image_value is out of line) */
    const unsigned char *texels;
    const float *noise;     /* mort_noise_texture records, 6152 B each */
    /* light sampling reads objects by their WORLD index (camera.light_obj_*):
     * world-order copies of every sphere / quad and of the hittable lists */
    const DSphere *wspheres; const DQuad *wquads;
    const int *list_types; const int *list_idxs; /* concatenated */
    /* per hittable_list: [MORT_NUM_HITTABLE_LIST] each.  Pointers into the blob, not arrays: a kernel's local view of the scene
     * must not hold a dynamically indexed member (it would live in private memory) */
    const int *list_first; const int *list_count;
    uint32_t blob_bytes;    /* size of everything above texels/noise (LDS staging candidate) */
    uint32_t lds_bytes;     /* bytes the staged part needs */
};

#endif
