/*
 * dev_math.h -- device-side vec3 / ray / XORWOW primitives for the gfx950
 * kernels.  Same operator semantics as the reference's vec3.cuh / ray.cuh /
 * rng.cuh (division = multiply by the fp32 reciprocal, dot sums left to
 * right, random_float = (float)(1.0 - curand_uniform)); compiled with
 * -ffp-contract=off so that nothing is fused (DESIGN.md "Numerical contract").
 */
#ifndef MORT_DEV_MATH_H
#define MORT_DEV_MATH_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mort_math.h"

#pragma clang fp contract(off)

#define DEV __host__ __device__ __forceinline__ /* the host loop (host_render.hip) runs the same bodies */

struct V3 { float x, y, z; };
DEV V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
DEV V3 vadd(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV V3 vsub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV V3 vmul(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
DEV V3 vneg(V3 a) { return mk(-a.x, -a.y, -a.z); }
DEV V3 vscale(float t, V3 v) { return mk(t * v.x, t * v.y, t * v.z); } /* vec3.cuh:99-101 */
DEV V3 vdiv(V3 v, float t) { return vscale(1 / t, v); }                /* vec3.cuh:109-112 */
DEV float vdot(V3 u, V3 v) { return u.x * v.x + u.y * v.y + u.z * v.z; }
DEV float vlen2(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
DEV float vlen(V3 a) { return mort_sqrtf(vlen2(a)); }
DEV V3 vunit(V3 a) { return vdiv(a, vlen(a)); }
DEV V3 vcross(V3 u, V3 v) { return mk(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x); }

struct Ray { V3 o, d; float tm; };
DEV V3 ray_at(const Ray &r, float t) { return vadd(r.o, vscale(t, r.d)); } /* ray.cuh:17-19 (t narrowed to fp32) */

/* ---- XORWOW in registers: the reference reloads and stores the 48-byte
 * state around every draw (rng.cuh:18-22); here a pixel's d, v[5] live in six
 * VGPRs from its first sample to its last. ---- */
struct Rng { uint32_t d, v0, v1, v2, v3, v4; uint32_t draws; };

DEV uint32_t xorwow_next(Rng &s) {
    uint32_t t = s.v0 ^ (s.v0 >> 2);
    s.v0 = s.v1; s.v1 = s.v2; s.v2 = s.v3; s.v3 = s.v4;
    s.v4 = (s.v4 ^ (s.v4 << 4)) ^ (t ^ (t << 1));
    s.d += 362437u;
    return s.v4 + s.d;
}
/* curand_uniform: x * 2^-32 + 2^-33 (the product is exact) */
DEV float curand_uniform_f(Rng &s) {
    s.draws++;
    uint32_t x = xorwow_next(s);
    return (float)x * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
}
/* rng.cuh:17-23 computes (float)(1.0 - (double)u).  u >= 2^-33 has 24 significant bits, so 1 - u is exact in fp64
 * unless u < 2^-29 (where both roundings give 1.0f); the fp32 subtraction rounds the same exact value once.
 * Checked for all 2^32 generator outputs (tests/test_rng.py). */
DEV float random_float(Rng &s) { return 1.0f - curand_uniform_f(s); }
DEV float random_float_range(Rng &s, float mn, float mx) { float b = random_float(s); return b * (mx - mn) + mn; }
DEV int random_int(Rng &s, int mn, int mx) { /* rng.cuh:31-42 */
    float random = curand_uniform_f(s);
    random = (float)((double)random * (mx - mn + 0.999999));
    random += (float)mn;
    return mort_f2i(__builtin_truncf(random));
}

DEV V3 random_in_unit_sphere(Rng &s) { /* vec3.cuh:143-155 */
    for (;;) {
        float a = random_float_range(s, -1, 1);
        float b = random_float_range(s, -1, 1);
        float c = random_float_range(s, -1, 1);
        V3 p = mk(a, b, c);
        if (vlen2(p) >= 1) continue;
        return p;
    }
}
DEV V3 random_unit_vector(Rng &s) { return vunit(random_in_unit_sphere(s)); }
DEV V3 random_in_unit_disk(Rng &s) { /* vec3.cuh:162-169 */
    for (;;) {
        float a = random_float_range(s, -1, 1);
        float b = random_float_range(s, -1, 1);
        V3 p = mk(a, b, 0);
        if (vlen2(p) < 1) return p;
    }
}
DEV V3 random_cosine_direction(Rng &s) { /* vec3.cuh:180-189 */
    float r1 = random_float(s);
    float r2 = random_float(s);
    float phi = (float)(2 * 3.1415926 * (double)r1);
    float sq = mort_sqrtf(r2);
    float sn, cs;
    mort_sincosf(phi, &sn, &cs); /* == mort_sinf(phi), mort_cosf(phi) */
    float x = cs * sq;
    float y = sn * sq;
    float z = mort_sqrtf(1 - r2);
    return mk(x, y, z);
}
DEV V3 reflect(V3 v, V3 n) { return vsub(v, vscale(2 * vdot(v, n), n)); } /* vec3.cuh:192-194 */
DEV V3 refract(V3 uv, V3 n, float etai_over_etat) { /* vec3.cuh:197-203 */
    double cos_theta = mort_fmin((double)vdot(vneg(uv), n), 1.0);
    V3 r_out_perp = vscale(etai_over_etat, vadd(uv, vscale((float)cos_theta, n)));
    V3 r_out_parallel = vscale((float)(-mort_sqrt(mort_fabs(1.0 - (double)vlen2(r_out_perp)))), n);
    return vadd(r_out_perp, r_out_parallel);
}
DEV float reflectance(float cosine, float ref_idx) { /* vec3.cuh:206-212 */
    float r0 = (1 - ref_idx) / (1 + ref_idx);
    r0 = r0 * r0;
    return r0 + (1 - r0) * mort_powi5f(1 - cosine);
}

struct Onb { V3 u, v, w; };
DEV Onb onb_from_w(V3 w) { /* onb.cuh:41-51 */
    Onb o;
    V3 unit_w = vunit(w);
    V3 a = ((double)mort_fabsf(unit_w.x) > 0.9) ? mk(0, 1, 0) : mk(1, 0, 0);
    V3 v = vunit(vcross(unit_w, a));
    V3 u = vcross(unit_w, v);
    o.u = u; o.v = v; o.w = unit_w;
    return o;
}
DEV V3 onb_local(const Onb &o, V3 a) { return vadd(vadd(vscale(a.x, o.u), vscale(a.y, o.v)), vscale(a.z, o.w)); }

#endif
