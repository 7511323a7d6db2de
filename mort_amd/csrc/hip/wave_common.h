/*
 * wave_common.h -- HBM records of MORT_MODE_WAVE, shared by its two forms (wave_bvh.h: reference-BVH worlds,
 * wave_gen.hip: every other world).  Front ray 32 B + id 4 B (two parities), hit 8 B (by position), pixel 48 B (by id:
 * XORWOW words, colour sum, packed counters), bounce stack [depth][id] 16 B.
 */
#ifndef MORT_WAVE_COMMON_H
#define MORT_WAVE_COMMON_H


#ifndef MORT_WF_SHADE_WAVES
#define MORT_WF_SHADE_WAVES 4 /* waves per SIMD the two shade kernels (wf_shade, wf_shade_gen) are compiled for: measured 4 / 5 / 6 / 8, DESIGN.md 4.5 */
#endif
#include "dev_render.h"

struct __attribute__((aligned(16))) WfRay { float ox, oy, oz, tm; float dx, dy, dz, time0; };
struct __attribute__((aligned(8))) WfHit { float t; int best; };
struct __attribute__((aligned(16))) WfPix {
    uint32_t d, v0, v1, v2;
    uint32_t v3, v4; float cr, cg;
    float cb; uint32_t packed; /* s_i | s_j << 12 | iter << 24 */ uint32_t segments, draws;
};

enum { WC_LAMB = 0, WC_SPEC = 1, WC_FIN = 2 };

struct WfCounters {
    unsigned front_count[2];  /* records in front[parity] */
    unsigned cls_count[2][3]; /* positions in the class queues of front parity */
    unsigned live;            /* pixels not finished yet */
    unsigned pad[7];
};

DEV Rng wf_rng_load(const WfPix &p) { Rng r; r.d = p.d; r.v0 = p.v0; r.v1 = p.v1; r.v2 = p.v2; r.v3 = p.v3; r.v4 = p.v4; r.draws = p.draws; return r; }
DEV void wf_rng_store(WfPix &p, const Rng &r) { p.d = r.d; p.v0 = r.v0; p.v1 = r.v1; p.v2 = r.v2; p.v3 = r.v3; p.v4 = r.v4; p.draws = r.draws; }


#endif
