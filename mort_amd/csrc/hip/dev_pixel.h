/*
 * dev_pixel.h -- Camera::render for ONE pixel (camera.cuh:179-208): the whole sample x bounce nest of
 * Camera::render / ray_color (camera.cuh:86-176) as one flat loop, so that a path that ends starts the pixel's next
 * sample at once.  RNG state in registers for the pixel's lifetime; bounce stack private; no global scratch arrays
 * (the reference allocates bounce_limit x W x H x 32 B, mort.cu:712-725).
 *
 * This one body is mega_kernel (one GPU lane per pixel, mort_hip.hip) AND the host loop of mort_hip_render_host
 * (host_render.hip: `mort --mode host`, the CPU figure the north_star asks to be timed beside the GPU).
 * TREE = false: world::hit over the flattened items (dev_trace.h world_hit); TREE = true: the single-lane walk of the
 * unified tree (dev_gen.h gen_world_hit).
 */
#ifndef MORT_DEV_PIXEL_H
#define MORT_DEV_PIXEL_H

#include "dev_gen.h"
#include "dev_shade.h"

#pragma clang fp contract(off)

struct PixelTotals { uint32_t segments, draws; };

/* lds_stack (device only): the first lds_levels bounce-stack levels of this lane live at lds_stack[level * lds_stride], the rest in the
 * private array (mega_kernel: 12 levels x 256 lanes x 16 B; the unwind reads them back one dependent level at a time, which from
 * private memory costs an L2 round trip per level).  The host loop passes none. */
template <bool TREE>
DEV PixelTotals render_pixel(const RenderArgs &a, const GenWalk *gw, int x, int ly, unsigned long long *scans, float4 *lds_stack = nullptr, int lds_levels = 0,
                             int lds_stride = 0) {
    const DScene &sc = a.sc;
    const int y = global_row(ly, a.rank, a.nranks, a.rows_per_block);
    const int lofs = x + ly * a.width;

    Rng rng;
    {
        const mort_rng_state st = a.states[lofs];
        rng.d = st.d; rng.v0 = st.v[0]; rng.v1 = st.v[1]; rng.v2 = st.v[2]; rng.v3 = st.v[3]; rng.v4 = st.v[4];
        rng.draws = 0;
    }

    StackEntry stack[MORT_MAX_BOUNCE_LIMIT];
    unsigned long long ident_mask = 0ull; /* levels whose entry is the identity (dielectric): not stored, see mega_bvh.h */
    V3 pixel_color = mk(0, 0, 0);
    const int spp = a.sqrt_spp * a.sqrt_spp;
    int s = 0, s_i = 0, s_j = 0;
    int iter = 0;
    bool fresh = true;
    uint32_t segments = 0;
    Ray ray;
    ray.o = mk(0, 0, 0); ray.d = mk(0, 0, 1); ray.tm = 0;
    float ray_time0 = 0.f;

    while (s < spp) {
        if (fresh) { /* camera.cuh:187-190 */
            ray = get_ray(a, x, y, rng, s_i, s_j);
            ray_time0 = ray.tm;
            iter = 0;
            fresh = false;
        }
        /* ---- one iteration of ray_color's bounce loop (camera.cuh:96-159) ---- */
        V3 final_value;
        bool done = false;
        if (iter >= a.bounce_limit) {
            final_value = mk(0, 0, 0);
            done = true;
        } else {
            Best best;
            segments++;
            const bool hit = TREE ? gen_world_hit(sc, *gw, ray, rng, best, scans) : world_hit(sc, ray, rng, best);
#ifdef MORT_DEBUG_PRINT
            if (lofs == a.debug_lofs) printf("[pix %d seg %u] o (%.9g %.9g %.9g) d (%.9g %.9g %.9g) tm %.9g -> hit %d t %.9g kind %d prim %d chain %d+%d draws %u\n", lofs, segments,
                ray.o.x, ray.o.y, ray.o.z, ray.d.x, ray.d.y, ray.d.z, ray.tm, (int)hit, hit ? best.t : 0.f, hit ? best.kind : 0, hit ? best.prim : 0, hit ? best.chain_first : 0, hit ? best.chain_count : 0, rng.draws);
#endif
            if (!hit) {
                final_value = a.background;
                done = true;
            } else {
                const ShadeOut so = shade_hit(sc, a.light_type, a.light_idx, ray, ray_time0, best, rng);
                if (so.done) { final_value = so.final_value; done = true; }
                else {
                    if (so.ident) ident_mask |= (1ull << iter);
                    else if (iter < lds_levels) { float4 e4; e4.x = so.e.kx; e4.y = so.e.ky; e4.z = so.e.kz; e4.w = so.e.rp; lds_stack[iter * lds_stride] = e4; }
                    else stack[iter] = so.e;
                    iter++;
                }
            }
        }
        if (done) { /* unwind (camera.cuh:165-173) and accumulate (camera.cuh:190) */
            while (iter > 0) {
                iter--;
                if ((ident_mask >> iter) & 1ull) { final_value = vadd(mk(0, 0, 0), final_value); continue; }
                StackEntry e;
                if (iter < lds_levels) { const float4 e4 = lds_stack[iter * lds_stride]; e.kx = e4.x; e.ky = e4.y; e.kz = e4.z; e.rp = e4.w; }
                else e = stack[iter];
                const V3 t = vmul(mk(e.kx, e.ky, e.kz), final_value);
                final_value = vadd(mk(0, 0, 0), vscale(e.rp, t));
            }
            ident_mask = 0ull;
            pixel_color = vadd(pixel_color, final_value);
            s++;
            s_i++;
            if (s_i == a.sqrt_spp) { s_i = 0; s_j++; }
            fresh = true;
        }
    }

    /* camera.cuh:194-207 */
    pixel_color = vscale(a.pixel_samples_scale, pixel_color);
    if (pixel_color.x != pixel_color.x) pixel_color.x = 0.0f;
    if (pixel_color.y != pixel_color.y) pixel_color.y = 0.0f;
    if (pixel_color.z != pixel_color.z) pixel_color.z = 0.0f;
    if (a.accum) { a.accum[3 * lofs] = pixel_color.x; a.accum[3 * lofs + 1] = pixel_color.y; a.accum[3 * lofs + 2] = pixel_color.z; }
    uchar4 out;
    {
        float c[3] = {mort_sqrtf(pixel_color.x), mort_sqrtf(pixel_color.y), mort_sqrtf(pixel_color.z)};
        unsigned char b[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float v = c[k];
            if (v < 0.0f) v = 0.0f;
            if (v > 0.999f) v = 0.999f;
            b[k] = (unsigned char)mort_f2i(256 * v);
        }
        out.x = b[0]; out.y = b[1]; out.z = b[2]; out.w = 255;
    }
    a.rgba[lofs] = out;
    if (a.seg_px) a.seg_px[lofs] = segments;
    {
        mort_rng_state st;
        st.d = rng.d; st.v[0] = rng.v0; st.v[1] = rng.v1; st.v[2] = rng.v2; st.v[3] = rng.v3; st.v[4] = rng.v4;
        st.boxmuller_flag = 0; st.boxmuller_flag_double = 0; st.boxmuller_extra = 0.f; st.boxmuller_extra_double = 0.;
        a.states[lofs] = st;
    }
    PixelTotals t; t.segments = segments; t.draws = rng.draws;
    return t;
}

#endif
