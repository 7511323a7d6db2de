/*
 * dev_render.h -- what every render kernel of libmort_hip.so shares: the launch arguments (camera, partition,
 * buffers), Camera::get_ray (camera.cuh:210-242), light-object sampling (pdf.cuh:60-80 over the objects.cuh
 * dispatchers) and the bounce-stack entry of ray_color's unwind (camera.cuh:165-173).
 */
#ifndef MORT_DEV_RENDER_H
#define MORT_DEV_RENDER_H

#include "mort_hip.h"
#include "dev_trace.h"

#pragma clang fp contract(off)

struct RenderArgs {
    DScene sc;
    /* camera (camera.cuh:13-45) */
    int width, height;
    int sqrt_spp, bounce_limit;
    float recip_sqrt_spp, pixel_samples_scale;
    V3 background, center, pixel00, du, dv, defocus_u, defocus_v;
    float defocus_angle;
    int light_type, light_idx;
    /* partition: this launch owns row blocks rank, rank + nranks, ... */
    int rank, nranks, rows_per_block, local_rows;
    /* buffers (packed owned rows) */
    mort_rng_state *states;
    uchar4 *rgba;
    float *accum;          /* may be null */
    uint32_t *seg_px;      /* may be null */
    unsigned long long *counters; /* [0] segments, [1] rng draws */
    int debug_lofs;        /* -DMORT_DEBUG_PRINT builds: packed pixel offset whose segments are printed (MORT_DEBUG_PIXEL), else unused */
};

DEV int global_row(int ly, int rank, int nranks, int rpb) {
    const int lb = ly / rpb, within = ly - lb * rpb;
    return (lb * nranks + rank) * rpb + within;
}

/* ---- Camera::get_ray (camera.cuh:210-242) ---- */
/* the camera fields get_ray reads, for kernels that keep them in LDS: their argument struct has a private copy (out-of-line helpers take
 * its address), and a new sample's ray otherwise starts with ten dependent loads from private memory */
struct CamView {
    float recip_sqrt_spp, defocus_angle;
    V3 center, pixel00, du, dv, defocus_u, defocus_v;
};
DEV void cam_view_fill(CamView &c, const RenderArgs &a) {
    c.recip_sqrt_spp = a.recip_sqrt_spp; c.defocus_angle = a.defocus_angle;
    c.center = a.center; c.pixel00 = a.pixel00; c.du = a.du; c.dv = a.dv; c.defocus_u = a.defocus_u; c.defocus_v = a.defocus_v;
}
template <typename CAM>
DEV Ray get_ray(const CAM &a, int x, int y, Rng &rng, int s_i, int s_j) {
    const double px = (double)(((float)s_i + random_float(rng)) * a.recip_sqrt_spp) - 0.5;
    const double py = (double)(((float)s_j + random_float(rng)) * a.recip_sqrt_spp) - 0.5;
    const float ox = (float)px, oy = (float)py;
    const V3 pixel_sample = vadd(vadd(a.pixel00, vscale((float)((double)x + (double)ox), a.du)),
                                 vscale((float)((double)y + (double)oy), a.dv));
    V3 origin;
    if (a.defocus_angle <= 0) {
        origin = a.center;
    } else {
        const V3 p = random_in_unit_disk(rng);
        origin = vadd(vadd(a.center, vscale(p.x, a.defocus_u)), vscale(p.y, a.defocus_v));
    }
    Ray r;
    r.o = origin;
    r.d = vsub(pixel_sample, origin);
    r.tm = random_float(rng);
    return r;
}

/* ---- light object sampling (pdf.cuh:60-80 over objects.cuh dispatchers) ---- */
DEV float light_pdf_value(const DScene &sc, int type, int idx, V3 origin, V3 direction) {
    if (type == MORT_OBJ_SPHERE) return wsphere_pdf_value(sc.wspheres[idx], origin, direction);
    if (type == MORT_OBJ_QUAD) return wquad_pdf_value(sc.wquads[idx], origin, direction);
    if (type == MORT_OBJ_HITTABLE_LIST) { /* objects.cuh:488-498 */
        const int first = sc.list_first[idx], n = sc.list_count[idx];
        const float weight = (float)(1.0 / (double)(float)n);
        float sum = 0.0f;
        for (int i = 0; i < n; i++) {
            const int t = sc.list_types[first + i], k = sc.list_idxs[first + i];
            float v = 0.0f;
            if (t == MORT_OBJ_SPHERE) v = wsphere_pdf_value(sc.wspheres[k], origin, direction);
            else if (t == MORT_OBJ_QUAD) v = wquad_pdf_value(sc.wquads[k], origin, direction);
            sum += weight * v;
        }
        return sum;
    }
    return 0.0f; /* pdfValueDispatch default (objects.cuh:961) */
}
DEV V3 light_random(const DScene &sc, int type, int idx, V3 origin, Rng &rng) {
    if (type == MORT_OBJ_HITTABLE_LIST) { /* objects.cuh:500-504 */
        const int first = sc.list_first[idx], n = sc.list_count[idx];
        const int k = random_int(rng, 0, n - 1);
        type = sc.list_types[first + k];
        idx = sc.list_idxs[first + k];
        if (type == MORT_OBJ_HITTABLE_LIST) return mk(1, 0, 0); /* nested lists are rejected on the host */
    }
    if (type == MORT_OBJ_SPHERE) return wsphere_random(sc.wspheres[idx], origin, rng);
    if (type == MORT_OBJ_QUAD) return wquad_random(sc.wquads[idx], origin, rng);
    return mk(1, 0, 0); /* randomDispatch default (objects.cuh:978) */
}

/* one bounce-stack entry: k = scattering_pdf * attenuation, rp = 1 / pdf.
 * The unwind (camera.cuh:166-173) computes emission + (1/pdf)*((spdf*att)*final);
 * spdf*att and 1/pdf are the same fp32 values whenever they are formed, and the
 * pushed emission is always (0,0,0): only diffuse_light emits and it never scatters. */
struct StackEntry { float kx, ky, kz, rp; };

#endif
