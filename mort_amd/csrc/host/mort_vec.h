/*
 * mort_vec.h -- fp32 vec3 helpers for the host scene layer, with the
 * reference's operator semantics (vec3.cuh:82-136): division by a scalar is
 * multiplication by the fp32 reciprocal; dot/length_squared sum left to right.
 */
#ifndef MORT_VEC_H
#define MORT_VEC_H

#include <math.h>
#include "mort_scene.h"

static inline mort_vec3 v3(float x, float y, float z) { mort_vec3 r = {{x, y, z}}; return r; }
static inline mort_vec3 v_add(mort_vec3 a, mort_vec3 b) { return v3(a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]); }
static inline mort_vec3 v_sub(mort_vec3 a, mort_vec3 b) { return v3(a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2]); }
static inline mort_vec3 v_mul(mort_vec3 a, mort_vec3 b) { return v3(a.e[0] * b.e[0], a.e[1] * b.e[1], a.e[2] * b.e[2]); }
static inline mort_vec3 v_neg(mort_vec3 a) { return v3(-a.e[0], -a.e[1], -a.e[2]); }
static inline mort_vec3 v_scale(float t, mort_vec3 v) { return v3(t * v.e[0], t * v.e[1], t * v.e[2]); }
static inline mort_vec3 v_div(mort_vec3 v, float t) { return v_scale(1 / t, v); }
static inline float v_dot(mort_vec3 a, mort_vec3 b) { return a.e[0] * b.e[0] + a.e[1] * b.e[1] + a.e[2] * b.e[2]; }
static inline float v_len2(mort_vec3 a) { return a.e[0] * a.e[0] + a.e[1] * a.e[1] + a.e[2] * a.e[2]; }
static inline float v_len(mort_vec3 a) { return sqrtf(v_len2(a)); }
static inline mort_vec3 v_unit(mort_vec3 a) { return v_div(a, v_len(a)); }
static inline mort_vec3 v_cross(mort_vec3 u, mort_vec3 v) {
    return v3(u.e[1] * v.e[2] - u.e[2] * v.e[1],
              u.e[2] * v.e[0] - u.e[0] * v.e[2],
              u.e[0] * v.e[1] - u.e[1] * v.e[0]);
}

#endif
