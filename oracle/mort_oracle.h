/*
 * mort_oracle.h -- CPU oracle for the render path.  TEST INFRASTRUCTURE ONLY
 * (see the header of mort_oracle.c): never linked into the product.
 */
#ifndef MORT_ORACLE_H
#define MORT_ORACLE_H

#include "mort_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

/* hit_record.cuh:10-19 */
typedef struct mort_oracle_hit {
    mort_vec3 p;
    mort_vec3 normal;
    int mat_idx;
    int mat_type;
    float t;
    float u, v;
    bool front_face;
} mort_oracle_hit;

typedef struct mort_oracle_stats {
    uint64_t segments;  /* world::hit calls */
    uint64_t rng_draws; /* curand_uniform calls */
} mort_oracle_stats;

/* curand_init(seed, subsequence, 0) */
void mort_oracle_rng_init(mort_rng_state *s, uint64_t seed, uint64_t subsequence);
/* setup_rng (rng.cuh:8-15): states[x + y*W] = curand_init(seed, x + y*W, 0) */
void mort_oracle_rng_seed(mort_rng_state *states, uint64_t seed, int width, int height);
unsigned mort_oracle_rng_next(mort_rng_state *s);
float mort_oracle_rng_uniform(mort_rng_state *s);          /* curand_uniform */
float mort_oracle_random_float(mort_rng_state *s);         /* rng.cuh:17-23 */
int mort_oracle_random_int(mort_rng_state *s, int mn, int mx); /* rng.cuh:31-42 */
void mort_oracle_rng_step_matrix_2p67(unsigned out[160 * 5]);

/* Camera::render (camera.cuh:178-208) over rows [row0,row1) with `nthreads`
 * host threads (rows interleaved).  rgba: W*H*4, accum: W*H*3 floats (linear
 * pixel mean after NaN scrub, before gamma) or NULL, segments_per_pixel: W*H
 * or NULL.  states advance exactly as the reference's would. */
int mort_oracle_render(const mort_world *w, const mort_camera *cam, mort_rng_state *states, int row0, int row1,
                       uint8_t *rgba, float *accum, uint32_t *segments_per_pixel, int nthreads,
                       mort_oracle_stats *stats);

/* known-answer entry points (ray7 = origin xyz, direction xyz, time) */
bool mort_oracle_world_hit(const mort_world *w, const float ray7[7], float t_min, float t_max,
                           mort_rng_state *state, mort_oracle_hit *out);
bool mort_oracle_aabb_hit(const mort_aabb *b, const float ray7[7], float t_min, float t_max);
void mort_oracle_get_ray(const mort_camera *cam, int x, int y, int s_i, int s_j, mort_rng_state *state, float ray7[7]);
void mort_oracle_ray_color(const mort_world *w, const mort_camera *cam, const float ray7[7], mort_rng_state *state, float rgb[3]);
void mort_oracle_texture_value(const mort_world *w, int tex_type, int tex_idx, float u, float v, const float p[3], float rgb[3]);
float mort_oracle_pdf_value(const mort_world *w, int type, int idx, const float origin[3], const float dir[3]);
void mort_oracle_light_random(const mort_world *w, int type, int idx, const float origin[3], mort_rng_state *state, float dir[3]);

float mort_oracle_sinf(float x);
float mort_oracle_cosf(float x);
void mort_oracle_sincosf(float x, float *s, float *c);
float mort_oracle_acosf(float x);
float mort_oracle_atan2f(float y, float x);
float mort_oracle_logf(float x);
double mort_oracle_sin(double x);

#ifdef __cplusplus
}
#endif
#endif
