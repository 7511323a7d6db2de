/*
 * wave_gen.h -- launch interface of wave_gen.hip: MORT_MODE_WAVE (the queued / wavefront form of the render path) for
 * worlds without reference BVHs -- reference scenes 2..9, i.e. BASELINE config 5's book-2 final scene.
 */
#ifndef MORT_WAVE_GEN_H
#define MORT_WAVE_GEN_H

#include "mega_gen.h"

struct WfGenHost { /* scratch owned by the context */
    void **d_wf; size_t *wf_bytes; unsigned **h_live; int *fronts; int num_cus;
};

/* Renders the owned rows through fronts of (path id, ray) records: wf_init, then one wf_trav_gen + wf_shade_gen launch
 * pair per front until no pixel is live.  ga: the unified-tree image and constants (as for mega_gen_kernel), ga.f.r the
 * camera / partition / buffers.  Returns hipSuccess, or hipErrorUnknown with *live_left > 0 when the front limit is hit. */
hipError_t mort_wave_gen_render(const GenArgs &ga, const WfGenHost &hb, int bounce_limit, int sqrt_spp, hipStream_t s, unsigned *live_left);
const void *mort_wave_gen_trav_kernel(bool prims_in_lds, int *block);

#endif
