/*
 * mort_ctx.h -- the opaque mort_ctx of include/mort_hip.h, shared by the translation units of libmort_hip.so.
 */
#ifndef MORT_CTX_H
#define MORT_CTX_H

#include <hip/hip_runtime.h>

#include <functional>
#include <string>
#include <vector>

#include "mort_hip.h"
#include "mega_gen.h"

struct mort_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string last_error;
    /* scene */
    void *d_scene = nullptr;
    DScene sc{};
    bool have_world = false;
    std::vector<int> list_types, list_idxs;        /* host copy, to validate the light object at render */
    int list_first[MORT_NUM_HITTABLE_LIST]{}, list_count[MORT_NUM_HITTABLE_LIST]{};
    int n_wspheres = 0, n_wquads = 0, n_lists = 0;
    /* partition */
    mort_partition part{0, 1, 8};
    /* rng */
    mort_rng_state *d_states = nullptr;
    int rng_w = 0, rng_h = 0, rng_local_rows = 0;
    uint32_t *d_seqmats = nullptr;
    /* MORT_MODE_THROUGHPUT: one stream per (pixel, stratum row), seeded on first use from the seed of mort_hip_rng_seed */
    uint64_t seed = 0;
    bool seed_known = false; /* false after rng_load: those states have no seed to derive sub-streams from */
    mort_rng_state *d_substates = nullptr;
    float *d_vaccum = nullptr;
    size_t substates_cap = 0, vaccum_cap = 0;
    int sub_w = 0, sub_h = 0, sub_lr = 0, sub_s = 0; /* what d_substates is seeded for (0 = nothing) */
    /* scratch */
    void *d_rgba = nullptr, *d_accum = nullptr, *d_segpx = nullptr;
    size_t rgba_cap = 0, accum_cap = 0, segpx_cap = 0;
    void *d_deep = nullptr; /* state-machine megakernels: bounce-stack levels below the LDS part, [level][lane of the launch] */
    size_t deep_cap = 0;
    void *d_wave_log = nullptr; /* profile builds, MORT_WAVE_LINES=1: 16 words per wave of the last state-machine launch */
    size_t wave_log_cap = 0, wave_log_waves = 0;
    unsigned long long *d_counters = nullptr; /* [0] segments, [1] rng draws, [2] work counter */
    bool wave_ok = false; /* wavefront mode: one BVH over spheres as the whole world, hot blob fits LDS */
    /* BVH megakernel: its own LDS image (four-wide own tree, leaf records with their spheres, leaf boxes, material / texture tables) */
    void *d_fast = nullptr;
    uint32_t f_lambert = 0, f_metal = 0, f_diel = 0, f_dlight = 0, f_iso = 0,
             f_solid = 0, f_checker = 0, fast_bytes = 0, f_nodes4 = 0, f_leafrecs = 0;
    void *d_trav = nullptr; /* wf_trav's LDS image */
    uint32_t t_nodes2 = 0, t_leaves = 0, t_spheres = 0, trav_bytes = 0;
    int own_nodes = 0, own_leaves = 0, own4_stack = 1;
    bool fast_ok = false;
    /* unified-tree megakernel (mega_gen.hip): its LDS image and launch constants */
    void *d_gen = nullptr;
    unsigned *d_prio_count = nullptr; int heavy_percent = 50; /* heavy waves: device word with the number of head tiles; its threshold */
    uint32_t gen_bytes = 0;
    GenArgs gen{};
    bool gen_ok = false;
    int gen_prims = 0; /* solid primitives in the unified tree */
    float gen_lo[3] = {0, 0, 0}, gen_hi[3] = {0, 0, 0}, gen_reach = 0;
    int num_cus = 256;
    /* pixel-tile ordering of the BVH megakernel: most expensive tiles first (cost = segments of the previous
     * frame with this geometry, or of a 1-sample probe) so the frame does not end on its longest pixel chains */
    unsigned *d_tile_cost = nullptr, *d_tile_order = nullptr;
    mort_rng_state *d_probe_states = nullptr;
    size_t tile_cap = 0, probe_cap = 0;
    unsigned long long cost_key = 0; /* hash of the (world, geometry, partition, camera basis) the costs in d_tile_cost belong to; 0 = none */
    unsigned world_serial = 0;       /* bumped by upload_world: part of cost_key */
    unsigned *d_tile_keys = nullptr, *d_tile_iota = nullptr; /* device argsort scratch (tile_sort.hip) */
    void *d_sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    hipStream_t last_stream = nullptr; /* stream of the most recent render launch (may be the caller's) */
    /* wavefront mode work buffers */
    void *d_wf = nullptr;
    size_t wf_bytes = 0;
    unsigned *h_live = nullptr; /* pinned */
    int wf_fronts = 0;          /* fronts of the last wavefront render (reported) */
    /* multi-GPU frame gather over RCCL (comm_rccl.hip): one communicator per context, rank = part.rank */
    void *comm = nullptr;            /* ncclComm_t */
    void *d_gather = nullptr;        /* rank 0: nranks x max tile bytes */
    void *d_frame = nullptr;         /* rank 0: the de-interleaved W x H x 4 frame */
    size_t gather_cap = 0, frame_cap = 0;
    hipEvent_t ev_g0 = nullptr, ev_g1 = nullptr; /* brackets of the gather step (created on first use) */
    /* mort_hip_render_gather: the render's statistics are collected after the gather's one host wait */
    bool defer_stats = false;
    std::function<int(mort_stats *)> pending_stats;
};

static inline int hip_fail(mort_ctx *c, hipError_t e, const char *what) {
    if (c) c->last_error = std::string(what) + ": " + hipGetErrorString(e);
    return MORT_ERR_HIP;
}
#define HIPCHK(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hip_fail(ctx, e_, #call); } while (0)


#endif
