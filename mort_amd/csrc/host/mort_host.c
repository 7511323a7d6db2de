/*
 * mort_host.c -- world containers, constructors, BVH builder, camera set-up.
 * See include/mort_host.h for the reference lines each group follows.
 * Built with -ffp-contract=off: every expression rounds exactly where the
 * reference's C++ expression does.
 */
#include "mort_host.h"
#include "mort_math.h"
#include "mort_vec.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ host RNG */

void mort_host_rng_init(mort_host_rng *g, uint32_t seed, int args_rtl) {
    g->state = seed;
    g->args_rtl = args_rtl;
}

/* MSVC CRT rand(): 32-bit LCG, 15 output bits (rng.cuh:44-58 sits on it). */
int mort_host_rand(mort_host_rng *g) {
    g->state = g->state * 214013u + 2531011u;
    return (int)((g->state >> 16) & 0x7fffu);
}

float mort_host_random_float(mort_host_rng *g) {
    return (float)(mort_host_rand(g) / (32767 + 1.0)); /* rng.cuh:46 */
}

float mort_host_random_float_range(mort_host_rng *g, float mn, float mx) {
    return mn + (mx - mn) * mort_host_random_float(g); /* rng.cuh:50-53 */
}

/* vec3::random() / vec3::random(min,max) (vec3.cuh:63-69): three sibling
 * arguments, evaluated in the profile's order. */
static mort_vec3 host_random_vec3(mort_host_rng *g) {
    float a = mort_host_random_float(g), b = mort_host_random_float(g), c = mort_host_random_float(g);
    return g->args_rtl ? v3(c, b, a) : v3(a, b, c);
}
static mort_vec3 host_random_vec3_range(mort_host_rng *g, float mn, float mx) {
    float a = mort_host_random_float_range(g, mn, mx);
    float b = mort_host_random_float_range(g, mn, mx);
    float c = mort_host_random_float_range(g, mn, mx);
    return g->args_rtl ? v3(c, b, a) : v3(a, b, c);
}
mort_vec3 mort_host_random_vec3(mort_host_rng *g) { return host_random_vec3(g); }
mort_vec3 mort_host_random_vec3_range(mort_host_rng *g, float mn, float mx) { return host_random_vec3_range(g, mn, mx); }

/* ------------------------------------------------------------------ aabb */

static mort_interval iv(float a, float b) { mort_interval r = {a, b}; return r; }
static mort_interval iv_union(mort_interval a, mort_interval b) {
    return iv(a.imin < b.imin ? a.imin : b.imin, a.imax > b.imax ? a.imax : b.imax); /* interval.cuh:12-15 */
}
static mort_aabb aabb_points(mort_vec3 a, mort_vec3 b) { /* aabb.cuh:17-21 */
    mort_aabb r;
    r.x = iv(fminf(a.e[0], b.e[0]), fmaxf(a.e[0], b.e[0]));
    r.y = iv(fminf(a.e[1], b.e[1]), fmaxf(a.e[1], b.e[1]));
    r.z = iv(fminf(a.e[2], b.e[2]), fmaxf(a.e[2], b.e[2]));
    return r;
}
static mort_aabb aabb_union(mort_aabb a, mort_aabb b) { /* aabb.cuh:24-28 */
    mort_aabb r;
    r.x = iv_union(a.x, b.x);
    r.y = iv_union(a.y, b.y);
    r.z = iv_union(a.z, b.z);
    return r;
}
static mort_aabb aabb_empty(void) { /* interval.cuh:46, aabb.cuh:73 */
    mort_aabb r;
    r.x = r.y = r.z = iv(HUGE_VALF, -HUGE_VALF);
    return r;
}
static mort_interval iv_shift(mort_interval i, double d) { /* interval.cuh:49-51 */
    return iv((float)(i.imin + d), (float)(i.imax + d));
}
static mort_aabb aabb_shift(mort_aabb b, mort_vec3 o) { /* aabb.cuh:76-79 */
    mort_aabb r;
    r.x = iv_shift(b.x, o.e[0]);
    r.y = iv_shift(b.y, o.e[1]);
    r.z = iv_shift(b.z, o.e[2]);
    return r;
}
static mort_interval aabb_axis(const mort_aabb *b, int n) { return n == 1 ? b->y : (n == 2 ? b->z : b->x); }
static int aabb_largest_axis(const mort_aabb *b) { /* aabb.cuh:62-67 */
    float sx = b->x.imax - b->x.imin, sy = b->y.imax - b->y.imin, sz = b->z.imax - b->z.imin;
    if (sx > sy) return sx > sz ? 0 : 2;
    return sy > sz ? 1 : 2;
}

mort_aabb mort_get_bbox(const mort_world *w, int t, int i) { /* objects.cuh:918-945 */
    switch (t) {
    case MORT_OBJ_SPHERE: return w->objs.host_sphere[i].bbox;
    case MORT_OBJ_QUAD: return w->objs.host_quad[i].bbox;
    case MORT_OBJ_TRANSLATE: return w->objs.host_translate[i].bbox;
    case MORT_OBJ_ROTATE_Y: return w->objs.host_rotate_y[i].bbox;
    case MORT_OBJ_CONSTANT_MEDIUM: return w->objs.host_constant_medium[i].bbox;
    case MORT_OBJ_HITTABLE_LIST: return w->objs.host_hittable_list[i].bbox;
    }
    return aabb_points(v3(0, 0, 0), v3(0, 0, 0));
}

/* ------------------------------------------------------------------ world */

int mort_world_init(mort_world *w) { /* world.cuh:20-25, objects.cuh:804-813 ... */
    memset(w, 0, sizeof *w);
    w->objs.host_sphere = calloc(MORT_NUM_SPHERES, sizeof(mort_sphere));
    w->objs.host_quad = calloc(MORT_NUM_QUADS, sizeof(mort_quad));
    w->objs.host_translate = calloc(MORT_NUM_TRANSLATE, sizeof(mort_translate));
    w->objs.host_rotate_y = calloc(MORT_NUM_ROTATE_Y, sizeof(mort_rotate_y));
    w->objs.host_constant_medium = calloc(MORT_NUM_CONSTANT_MEDIUM, sizeof(mort_constant_medium));
    w->objs.host_hittable_list = calloc(MORT_NUM_HITTABLE_LIST, sizeof(mort_hittable_list));
    w->objs.host_bvh = calloc(MORT_NUM_BVH, sizeof(mort_bvh));
    w->mats.host_lambertian = calloc(MORT_NUM_LAMBERTIANS, sizeof(mort_lambertian));
    w->mats.host_metal = calloc(MORT_NUM_METALS, sizeof(mort_metal));
    w->mats.host_dielectric = calloc(MORT_NUM_DIELECTRICS, sizeof(mort_dielectric));
    w->mats.host_diffuse_light = calloc(MORT_NUM_DIFFUSE_LIGHTS, sizeof(mort_diffuse_light));
    w->mats.host_isotropic = calloc(MORT_NUM_ISOTROPICS, sizeof(mort_isotropic));
    w->texs.host_solid_color = calloc(MORT_NUM_SOLID_COLOR, sizeof(mort_solid_color));
    w->texs.host_checker_texture = calloc(MORT_NUM_CHECKER_TEXTURE, sizeof(mort_checker_texture));
    w->texs.host_image_texture = calloc(MORT_NUM_IMAGE_TEXTURE, sizeof(mort_image_texture));
    w->texs.host_noise_texture = calloc(MORT_NUM_NOISE_TEXTURE, sizeof(mort_noise_texture));
    w->bvh_mode = false;
    if (!w->objs.host_sphere || !w->objs.host_quad || !w->objs.host_translate || !w->objs.host_rotate_y ||
        !w->objs.host_constant_medium || !w->objs.host_hittable_list || !w->objs.host_bvh ||
        !w->mats.host_lambertian || !w->mats.host_metal || !w->mats.host_dielectric ||
        !w->mats.host_diffuse_light || !w->mats.host_isotropic || !w->texs.host_solid_color ||
        !w->texs.host_checker_texture || !w->texs.host_image_texture || !w->texs.host_noise_texture) {
        mort_world_free(w);
        return -1;
    }
    return 0;
}

void mort_world_free(mort_world *w) {
    free(w->objs.host_sphere); free(w->objs.host_quad); free(w->objs.host_translate);
    free(w->objs.host_rotate_y); free(w->objs.host_constant_medium);
    free(w->objs.host_hittable_list); free(w->objs.host_bvh);
    free(w->mats.host_lambertian); free(w->mats.host_metal); free(w->mats.host_dielectric);
    free(w->mats.host_diffuse_light); free(w->mats.host_isotropic);
    free(w->texs.host_solid_color); free(w->texs.host_checker_texture);
    free(w->texs.host_image_texture); free(w->texs.host_noise_texture);
    memset(w, 0, sizeof *w);
}

/* ------------------------------------------------------------------ textures */

int mort_add_solid_color(mort_world *w, mort_vec3 c) {
    if (w->texs.num_solid_colors >= MORT_NUM_SOLID_COLOR) return -1;
    int i = w->texs.num_solid_colors++;
    w->texs.host_solid_color[i].color_value = c;
    w->texs.host_solid_color[i].idx = i;
    return i;
}

int mort_add_checker_texture(mort_world *w, float scale, int et, int ei, int ot, int oi) {
    if (w->texs.num_checker_textures >= MORT_NUM_CHECKER_TEXTURE) return -1;
    int i = w->texs.num_checker_textures++;
    mort_checker_texture *t = &w->texs.host_checker_texture[i];
    t->inv_scale = (float)(1.0 / scale); /* textures.cuh:43 */
    t->evenTextureType = et; t->evenTextureIdx = ei;
    t->oddTextureType = ot; t->oddTextureIdx = oi;
    t->idx = i;
    return i;
}

int mort_add_image_texture(mort_world *w, const unsigned char *texels, int width, int height) {
    if (w->texs.num_image_textures >= MORT_NUM_IMAGE_TEXTURE) return -1;
    int i = w->texs.num_image_textures++;
    mort_image_texture *t = &w->texs.host_image_texture[i];
    t->texels = texels;
    t->width = texels ? width : 0;   /* img_loader.h:46-47: a failed load reports 0x0 */
    t->height = texels ? height : 0;
    t->idx = i;
    return i;
}

static void perlin_generate_perm(int *arr, mort_host_rng *g) { /* textures.cuh:216-230 */
    for (int i = 0; i < MORT_POINT_COUNT; i++) arr[i] = i;
    for (int i = MORT_POINT_COUNT - 1; i > 0; i--) {
        int target = (int)mort_host_random_float_range(g, 0.0f, (float)i);
        int tmp = arr[i];
        arr[i] = arr[target];
        arr[target] = tmp;
    }
}

int mort_add_noise_texture(mort_world *w, float scale, mort_host_rng *g) { /* textures.cuh:164-172 */
    if (w->texs.num_noise_textures >= MORT_NUM_NOISE_TEXTURE) return -1;
    int i = w->texs.num_noise_textures++;
    mort_noise_texture *t = &w->texs.host_noise_texture[i];
    t->scale = scale;
    for (int k = 0; k < MORT_POINT_COUNT; ++k)
        t->ranvec[k] = v_unit(host_random_vec3_range(g, -1, 1));
    perlin_generate_perm(t->perm_x, g);
    perlin_generate_perm(t->perm_y, g);
    perlin_generate_perm(t->perm_z, g);
    t->idx = i; /* the reference leaves idx unset; capacity is 1, so 0 */
    return i;
}

/* ------------------------------------------------------------------ materials */

int mort_add_lambertian(mort_world *w, int tt, int ti) {
    if (w->mats.num_lambertians >= MORT_NUM_LAMBERTIANS) return -1;
    int i = w->mats.num_lambertians++;
    mort_lambertian m = {tt, ti, i};
    w->mats.host_lambertian[i] = m;
    return i;
}
int mort_add_metal(mort_world *w, mort_vec3 albedo, float fuzz) {
    if (w->mats.num_metals >= MORT_NUM_METALS) return -1;
    int i = w->mats.num_metals++;
    mort_metal m = {albedo, fuzz, i};
    w->mats.host_metal[i] = m;
    return i;
}
int mort_add_dielectric(mort_world *w, float ri) { /* materials.cuh:105 */
    if (w->mats.num_dielectrics >= MORT_NUM_DIELECTRICS) return -1;
    int i = w->mats.num_dielectrics++;
    mort_dielectric m = {ri, (float)(1.0 / ri), {{1.0f, 1.0f, 1.0f}}, i};
    w->mats.host_dielectric[i] = m;
    return i;
}
int mort_add_diffuse_light(mort_world *w, int tt, int ti) {
    if (w->mats.num_diffuse_lights >= MORT_NUM_DIFFUSE_LIGHTS) return -1;
    int i = w->mats.num_diffuse_lights++;
    mort_diffuse_light m = {tt, ti, i};
    w->mats.host_diffuse_light[i] = m;
    return i;
}
int mort_add_isotropic(mort_world *w, int tt, int ti) {
    if (w->mats.num_isotropics >= MORT_NUM_ISOTROPICS) return -1;
    int i = w->mats.num_isotropics++;
    mort_isotropic m = {tt, ti, i};
    w->mats.host_isotropic[i] = m;
    return i;
}

/* ------------------------------------------------------------------ objects */

int mort_add_sphere(mort_world *w, mort_vec3 cen, float r, int mt, int mi, bool skip) { /* objects.cuh:38-43 */
    if (w->objs.num_spheres >= MORT_NUM_SPHERES) return -1;
    int i = w->objs.num_spheres++;
    mort_sphere *s = &w->objs.host_sphere[i];
    memset(s, 0, sizeof *s);
    s->center1 = cen; s->radius = r; s->moves = false;
    s->mat_type = mt; s->mat_idx = mi; s->idx = i; s->skip = skip;
    mort_vec3 rv = v3(r, r, r);
    s->bbox = aabb_points(v_sub(cen, rv), v_add(cen, rv));
    return i;
}

int mort_add_moving_sphere(mort_world *w, mort_vec3 c1, mort_vec3 c2, float r, int mt, int mi, bool skip) {
    if (w->objs.num_spheres >= MORT_NUM_SPHERES) return -1; /* objects.cuh:46-55 */
    int i = w->objs.num_spheres++;
    mort_sphere *s = &w->objs.host_sphere[i];
    memset(s, 0, sizeof *s);
    s->center1 = c1; s->radius = r; s->moves = true;
    s->center_vec = v_sub(c2, c1);
    s->mat_type = mt; s->mat_idx = mi; s->idx = i; s->skip = skip;
    mort_vec3 rv = v3(r, r, r);
    s->bbox = aabb_union(aabb_points(v_sub(c1, rv), v_add(c1, rv)), aabb_points(v_sub(c2, rv), v_add(c2, rv)));
    return i;
}

int mort_add_quad(mort_world *w, mort_vec3 Q, mort_vec3 u, mort_vec3 v, int mt, int mi, bool skip) {
    if (w->objs.num_quads >= MORT_NUM_QUADS) return -1; /* objects.cuh:170-185 */
    int i = w->objs.num_quads++;
    mort_quad *q = &w->objs.host_quad[i];
    memset(q, 0, sizeof *q);
    q->Q = Q; q->u = u; q->v = v; q->mat_type = mt; q->mat_idx = mi;
    mort_vec3 n = v_cross(u, v);
    q->normal = v_unit(n);
    q->D = v_dot(q->normal, Q);
    q->w = v_div(n, v_dot(n, n));
    q->area = v_len(n);
    q->idx = i; q->skip = skip;
    mort_aabb d1 = aabb_points(Q, v_add(v_add(Q, u), v));
    mort_aabb d2 = aabb_points(v_add(Q, u), v_add(Q, v));
    q->bbox = aabb_union(d1, d2);
    return i;
}

int mort_add_translate(mort_world *w, int ot, int oi, mort_vec3 d, bool skip) { /* objects.cuh:258-263 */
    if (w->objs.num_translates >= MORT_NUM_TRANSLATE) return -1;
    int i = w->objs.num_translates++;
    mort_translate *t = &w->objs.host_translate[i];
    memset(t, 0, sizeof *t);
    t->obj_type = ot; t->obj_idx = oi; t->offset = d; t->idx = i; t->skip = skip;
    t->bbox = aabb_shift(mort_get_bbox(w, ot, oi), d);
    return i;
}

int mort_add_rotate_y(mort_world *w, int ot, int oi, float theta, bool skip) { /* objects.cuh:296-329 */
    if (w->objs.num_rotate_y >= MORT_NUM_ROTATE_Y) return -1;
    int i = w->objs.num_rotate_y++;
    mort_rotate_y *r = &w->objs.host_rotate_y[i];
    memset(r, 0, sizeof *r);
    r->obj_type = ot; r->obj_idx = oi; r->idx = i; r->skip = skip;
    float radians = (float)(theta * 3.1415926535897932385 / 180.0);
    r->sin_theta = mort_sinf(radians);
    r->cos_theta = mort_cosf(radians);
    mort_aabb b = mort_get_bbox(w, ot, oi);
    float pmin[3] = {HUGE_VALF, HUGE_VALF, HUGE_VALF}, pmax[3] = {-HUGE_VALF, -HUGE_VALF, -HUGE_VALF};
    for (int a = 0; a < 2; a++)
        for (int bb = 0; bb < 2; bb++)
            for (int c = 0; c < 2; c++) {
                float x = a * b.x.imax + (1 - a) * b.x.imin;
                float y = bb * b.y.imax + (1 - bb) * b.y.imin;
                float z = c * b.z.imax + (1 - c) * b.z.imin;
                float nx = r->cos_theta * x + r->sin_theta * z;
                float nz = -r->sin_theta * x + r->cos_theta * z;
                float t[3] = {nx, y, nz};
                for (int k = 0; k < 3; k++) {
                    pmin[k] = fminf(pmin[k], t[k]);
                    pmax[k] = fmaxf(pmax[k], t[k]);
                }
            }
    r->bbox = aabb_points(v3(pmin[0], pmin[1], pmin[2]), v3(pmax[0], pmax[1], pmax[2]));
    return i;
}

int mort_add_constant_medium(mort_world *w, int ot, int oi, float d, int mt, int mi, bool skip) {
    if (w->objs.num_constant_medium >= MORT_NUM_CONSTANT_MEDIUM) return -1; /* objects.cuh:384-394 */
    int i = w->objs.num_constant_medium++;
    mort_constant_medium *m = &w->objs.host_constant_medium[i];
    memset(m, 0, sizeof *m);
    m->obj_type = ot; m->obj_idx = oi;
    m->neg_inv_density = -(1.0 / d);
    m->mat_type = mt; m->mat_idx = mi; m->idx = i; m->skip = skip;
    m->bbox = mort_get_bbox(w, ot, oi);
    return i;
}

int mort_add_hittable_list(mort_world *w, bool skip) { /* objects.cuh:459 */
    if (w->objs.num_hittable_list >= MORT_NUM_HITTABLE_LIST) return -1;
    int i = w->objs.num_hittable_list++;
    mort_hittable_list *l = &w->objs.host_hittable_list[i];
    memset(l, 0, sizeof *l);
    l->skip = skip; l->idx = i; l->num_objs = 0;
    l->bbox = aabb_points(v3(0, 0, 0), v3(0, 0, 0));
    return i;
}

int mort_list_add(mort_world *w, int li, int ot, int oi) { /* objects.cuh:462-469 */
    if (li < 0 || li >= w->objs.num_hittable_list) return -1;
    mort_hittable_list *l = &w->objs.host_hittable_list[li];
    if (l->num_objs >= MORT_LIST_MAX_OBJS) return -1; /* the reference drops silently */
    l->obj_types[l->num_objs] = ot;
    l->obj_idxs[l->num_objs] = oi;
    mort_aabb b = mort_get_bbox(w, ot, oi);
    l->bbox = (l->num_objs == 0) ? b : aabb_union(l->bbox, b);
    l->num_objs += 1;
    return 0;
}

/* ------------------------------------------------------------------ BVH build */

static int compare_by_axis(const mort_world *w, const mort_hittable_list *l, int a, int b, int axis) {
    /* objects.cuh:981-1000: order by bbox minimum on the axis */
    mort_aabb ba = mort_get_bbox(w, l->obj_types[a], l->obj_idxs[a]);
    mort_aabb bb = mort_get_bbox(w, l->obj_types[b], l->obj_idxs[b]);
    float ma = aabb_axis(&ba, axis).imin, mb = aabb_axis(&bb, axis).imin;
    if (ma < mb) return -1;
    if (ma > mb) return 1;
    return 0;
}

#define SWAP_T(T, arr, i, j) do { T tmp_ = (arr)[i]; (arr)[i] = (arr)[j]; (arr)[j] = tmp_; } while (0)

static void swap_objects(mort_world *w, int t, int i, int j) { /* objects.cuh:815-845 */
    switch (t) {
    case MORT_OBJ_SPHERE: SWAP_T(mort_sphere, w->objs.host_sphere, i, j); break;
    case MORT_OBJ_QUAD: SWAP_T(mort_quad, w->objs.host_quad, i, j); break;
    case MORT_OBJ_TRANSLATE: SWAP_T(mort_translate, w->objs.host_translate, i, j); break;
    case MORT_OBJ_ROTATE_Y: SWAP_T(mort_rotate_y, w->objs.host_rotate_y, i, j); break;
    case MORT_OBJ_CONSTANT_MEDIUM: SWAP_T(mort_constant_medium, w->objs.host_constant_medium, i, j); break;
    default: break; /* lists / bvhs are 8-42 KB; the built-in scenes never sort them */
    }
}

/* The reference bubble-sorts each span (objects.cuh:631-661), swapping only
 * on a strict "greater"; that is a stable sort, so a stable insertion sort
 * over the same comparison and the same swap primitive leaves the same order
 * (and, for same-type neighbours, the same physical object placement). */
static void sort_span(mort_world *w, mort_hittable_list *l, int start, int end, int axis) {
    for (int i = start + 1; i < end; i++) {
        for (int j = i; j > start && compare_by_axis(w, l, j - 1, j, axis) == 1; j--) {
            if (l->obj_types[j - 1] == l->obj_types[j]) {
                swap_objects(w, l->obj_types[j], l->obj_idxs[j - 1], l->obj_idxs[j]);
            } else {
                SWAP_T(int, l->obj_types, j - 1, j);
                SWAP_T(int, l->obj_idxs, j - 1, j);
            }
        }
    }
}

int mort_add_bvh(mort_world *w, int li, bool skip) { /* objects.cuh:529-611 */
    if (w->objs.num_bvh >= MORT_NUM_BVH) return -1;
    if (li < 0 || li >= w->objs.num_hittable_list) return -1;
    mort_hittable_list *l = &w->objs.host_hittable_list[li];
    int bi = w->objs.num_bvh;
    mort_bvh *b = &w->objs.host_bvh[bi];
    memset(b, 0, sizeof *b);
    b->idx = bi; b->skip = skip;

    static int span_begin[MORT_MAX_BVH_NODES], span_end[MORT_MAX_BVH_NODES];
    int size = 1, cur = 0;
    span_begin[0] = 0;
    span_end[0] = l->num_objs;
    while (cur < size) {
        int s = span_begin[cur], e = span_end[cur];
        b->bounding_boxes[cur] = aabb_empty();
        for (int i = s; i < e; i++)
            b->bounding_boxes[cur] = aabb_union(b->bounding_boxes[cur], mort_get_bbox(w, l->obj_types[i], l->obj_idxs[i]));
        int axis = aabb_largest_axis(&b->bounding_boxes[cur]);
        int span = e - s;
        if (span == 1) {
            b->left_children_types[cur] = b->right_children_types[cur] = l->obj_types[s];
            b->left_children_idxs[cur] = b->right_children_idxs[cur] = l->obj_idxs[s];
            b->is_internal_node[cur] = false;
        } else if (span == 2) {
            int lo = s, hi = s + 1;
            if (compare_by_axis(w, l, s, s + 1, axis) > 0) { lo = s + 1; hi = s; }
            b->left_children_types[cur] = l->obj_types[lo];
            b->left_children_idxs[cur] = l->obj_idxs[lo];
            b->right_children_types[cur] = l->obj_types[hi];
            b->right_children_idxs[cur] = l->obj_idxs[hi];
            b->is_internal_node[cur] = false;
        } else {
            if (size + 2 > MORT_MAX_BVH_NODES) return -1;
            sort_span(w, l, s, e, axis);
            int mid = s + (span / 2 + (span % 2 != 0));
            b->left_children_types[cur] = MORT_OBJ_BVH;
            b->left_children_idxs[cur] = size;
            span_begin[size] = s; span_end[size] = mid; size++;
            b->right_children_types[cur] = MORT_OBJ_BVH;
            b->right_children_idxs[cur] = size;
            span_begin[size] = mid; span_end[size] = e; size++;
            b->is_internal_node[cur] = true;
        }
        cur++;
    }
    w->objs.num_bvh++;
    w->bvh_mode = true; /* world.cuh:51-54 */
    return bi;
}

/* ------------------------------------------------------------------ box helpers */

void mort_box(mort_world *w, mort_vec3 a, mort_vec3 b, int mt, int mi) { /* utils.h:51-67 */
    mort_vec3 mn = v3(fminf(a.e[0], b.e[0]), fminf(a.e[1], b.e[1]), fminf(a.e[2], b.e[2]));
    mort_vec3 mx = v3(fmaxf(a.e[0], b.e[0]), fmaxf(a.e[1], b.e[1]), fmaxf(a.e[2], b.e[2]));
    mort_vec3 dx = v3(mx.e[0] - mn.e[0], 0, 0);
    mort_vec3 dy = v3(0, mx.e[1] - mn.e[1], 0);
    mort_vec3 dz = v3(0, 0, mx.e[2] - mn.e[2]);
    mort_add_quad(w, v3(mn.e[0], mn.e[1], mx.e[2]), dx, dy, mt, mi, false);          /* front */
    mort_add_quad(w, v3(mx.e[0], mn.e[1], mx.e[2]), v_neg(dz), dy, mt, mi, false);   /* right */
    mort_add_quad(w, v3(mx.e[0], mn.e[1], mn.e[2]), v_neg(dx), dy, mt, mi, false);   /* back */
    mort_add_quad(w, v3(mn.e[0], mn.e[1], mn.e[2]), dz, dy, mt, mi, false);          /* left */
    mort_add_quad(w, v3(mn.e[0], mx.e[1], mx.e[2]), dx, v_neg(dz), mt, mi, false);   /* top */
    mort_add_quad(w, v3(mn.e[0], mn.e[1], mn.e[2]), dx, dz, mt, mi, false);          /* bottom */
}

/* six skip-quads -> skip-list -> skip rotate_y -> translate (utils.h:69-96).
 * Returns the translate's idx. */
static int rotated_box_common(mort_world *w, mort_vec3 size, mort_vec3 tr, float theta, int mt, int mi, bool tr_skip) {
    mort_vec3 dx = v3(size.e[0], 0, 0), dy = v3(0, size.e[1], 0), dz = v3(0, 0, size.e[2]);
    int q[6];
    q[0] = mort_add_quad(w, v3(0, 0, size.e[2]), dx, dy, mt, mi, true);
    q[1] = mort_add_quad(w, v3(size.e[0], 0, size.e[2]), v_neg(dz), dy, mt, mi, true);
    q[2] = mort_add_quad(w, v3(size.e[0], 0, 0), v_neg(dx), dy, mt, mi, true);
    q[3] = mort_add_quad(w, v3(0, 0, 0), dz, dy, mt, mi, true);
    q[4] = mort_add_quad(w, v3(0, size.e[1], size.e[2]), dx, v_neg(dz), mt, mi, true);
    q[5] = mort_add_quad(w, v3(0, 0, 0), dx, dz, mt, mi, true);
    int l = mort_add_hittable_list(w, true);
    for (int k = 0; k < 6; k++) mort_list_add(w, l, MORT_OBJ_QUAD, q[k]);
    int r = mort_add_rotate_y(w, MORT_OBJ_HITTABLE_LIST, l, theta, true);
    return mort_add_translate(w, MORT_OBJ_ROTATE_Y, r, tr, tr_skip);
}

void mort_rotated_box(mort_world *w, mort_vec3 size, mort_vec3 tr, float theta, int mt, int mi) {
    rotated_box_common(w, size, tr, theta, mt, mi, false);
}

void mort_rotated_smoke_box(mort_world *w, mort_vec3 size, mort_vec3 tr, float theta, float d, int mt, int mi) {
    int t = rotated_box_common(w, size, tr, theta, mt, mi, true); /* utils.h:98-126 */
    mort_add_constant_medium(w, MORT_OBJ_TRANSLATE, t, d, mt, mi, false);
}

/* ------------------------------------------------------------------ camera */

void mort_camera_defaults(mort_camera *c) { /* camera.cuh:13-43 */
    memset(c, 0, sizeof *c);
    c->aspect_ratio = 1.0f;
    c->image_width = 1500;
    c->samples_per_pixel = 50;
    c->bounce_limit = 10;
    c->vfov = 90;
    c->background = v3(0.70f, 0.80f, 1.00f);
    c->light_obj_type = -1; /* uninitialised in the reference; every scene sets it */
    c->light_obj_idx = 0;
    c->lookfrom = v3(0, 0, 1);
    c->lookat = v3(0, 0, 0);
    c->vup = v3(0, 1, 0);
    c->defocus_angle = 0;
    c->focus_dist = 10;
}

static float degrees_to_radians(float degrees) { /* utils.h:21,25-27: float pi */
    const float pi = 3.1415926535897932385f;
    return (float)(degrees * pi / 180.0);
}
static float host_tanf(float x) { return (float)(mort_sin((double)x) / mort_cos((double)x)); }

void mort_camera_initialize(mort_camera *c);

/* rotate_around (vec3.cuh:215-227): `vec` turned by theta about `axis` */
static mort_vec3 rotate_around(mort_vec3 vec, mort_vec3 axis, float theta) {
    const mort_vec3 a_parallel_b = v_scale(v_dot(vec, axis) / v_dot(axis, axis), axis);
    const mort_vec3 a_orthogonal_b = v_sub(vec, a_parallel_b);
    const mort_vec3 w = v_cross(axis, a_orthogonal_b);
    const float x1 = (float)mort_cos((double)theta) / v_len(a_orthogonal_b);
    const float x2 = (float)mort_sin((double)theta) / v_len(w);
    const mort_vec3 rot = v_scale(v_len(a_orthogonal_b), v_add(v_scale(x1, a_orthogonal_b), v_scale(x2, w)));
    return v_add(rot, a_parallel_b);
}

/* One idle tick of the reference's input() (mort.cu:49-91), with the Win32 polling replaced by its result: which of W/S/A/D are
 * down (MORT_KEY_*), the mouse delta in pixels, and whether the left button is held.  Moves lookfrom / lookat along the camera
 * basis of the PREVIOUS initialize(), turns lookat about vup / u by -delta / 500 rad, then re-initialises the camera. */
void mort_camera_input(mort_camera *c, int keys, int mouse_dx, int mouse_dy, int left_button) {
    if (keys & MORT_KEY_W) { c->lookat = v_add(c->lookat, v_neg(c->w)); c->lookfrom = v_add(c->lookfrom, v_neg(c->w)); }
    if (keys & MORT_KEY_S) { c->lookat = v_add(c->lookat, c->w); c->lookfrom = v_add(c->lookfrom, c->w); }
    if (keys & MORT_KEY_A) { c->lookat = v_add(c->lookat, v_neg(c->u)); c->lookfrom = v_add(c->lookfrom, v_neg(c->u)); }
    if (keys & MORT_KEY_D) { c->lookat = v_add(c->lookat, c->u); c->lookfrom = v_add(c->lookfrom, c->u); }
    if (left_button) {
        if (mouse_dx != 0) {
            const mort_vec3 dir = v_sub(c->lookat, c->lookfrom);
            c->lookat = v_add(c->lookfrom, rotate_around(dir, c->vup, (float)(-mouse_dx / 500.0)));
        }
        if (mouse_dy != 0) {
            const mort_vec3 dir = v_sub(c->lookat, c->lookfrom);
            c->lookat = v_add(c->lookfrom, rotate_around(dir, c->u, (float)(-mouse_dy / 500.0)));
        }
    }
    mort_camera_initialize(c);
}

void mort_camera_initialize(mort_camera *c) { /* camera.cuh:47-84 */
    c->image_height = (int)(c->image_width / c->aspect_ratio);
    c->image_height = (c->image_height < 1) ? 1 : c->image_height;

    c->sqrt_spp = (int)sqrt((double)c->samples_per_pixel);
    c->pixel_samples_scale = (float)(1.0 / (c->sqrt_spp * c->sqrt_spp));
    c->recip_sqrt_spp = (float)(1.0 / c->sqrt_spp);

    c->center = c->lookfrom;

    float theta = degrees_to_radians((float)c->vfov);
    float h = host_tanf(theta / 2);
    float viewport_height = 2 * h * c->focus_dist;
    double viewport_width = viewport_height * ((double)c->image_width / c->image_height);

    c->w = v_unit(v_sub(c->lookfrom, c->lookat));
    c->u = v_unit(v_cross(c->vup, c->w));
    c->v = v_cross(c->w, c->u);

    mort_vec3 viewport_u = v_scale((float)viewport_width, c->u);
    mort_vec3 viewport_v = v_scale(viewport_height, v_neg(c->v));

    c->pixel_delta_u = v_div(viewport_u, (float)c->image_width);
    c->pixel_delta_v = v_div(v_neg(viewport_v), (float)c->image_height);

    mort_vec3 ul = v_add(v_sub(v_sub(c->center, v_scale(c->focus_dist, c->w)), v_div(viewport_u, 2)), v_div(viewport_v, 2));
    c->pixel00_loc = v_add(ul, v_scale(0.5f, v_add(c->pixel_delta_u, c->pixel_delta_v)));

    float defocus_radius = c->focus_dist * host_tanf(degrees_to_radians(c->defocus_angle / 2));
    c->defocus_disk_u = v_scale(defocus_radius, c->u);
    c->defocus_disk_v = v_scale(defocus_radius, c->v);
}

int mort_camera_effective_spp(const mort_camera *c) {
    int s = (int)sqrt((double)c->samples_per_pixel);
    return s * s;
}

/* ------------------------------------------------------------------ PPM */

int mort_write_ppm(const char *path, const uint8_t *rgba, int width, int height) {
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    fprintf(f, "P6\n%d %d\n255\n", width, height);
    for (int y = height - 1; y >= 0; y--)
        for (int x = 0; x < width; x++)
            fwrite(rgba + ((size_t)y * width + x) * 4, 1, 3, f);
    return fclose(f);
}

unsigned char *mort_read_ppm(const char *path, int *width, int *height) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    int w = 0, h = 0, mx = 0;
    char magic[3] = {0};
    if (fscanf(f, "%2s", magic) != 1 || strcmp(magic, "P6") != 0) { fclose(f); return NULL; }
    int vals[3], n = 0;
    while (n < 3) {
        int ch = fgetc(f);
        if (ch == EOF) { fclose(f); return NULL; }
        if (ch == '#') { while (ch != '\n' && ch != EOF) ch = fgetc(f); continue; }
        if (ch == ' ' || ch == '\n' || ch == '\r' || ch == '\t') continue;
        ungetc(ch, f);
        if (fscanf(f, "%d", &vals[n]) != 1) { fclose(f); return NULL; }
        n++;
    }
    fgetc(f); /* single whitespace after maxval */
    w = vals[0]; h = vals[1]; mx = vals[2];
    if (w <= 0 || h <= 0 || mx != 255) { fclose(f); return NULL; }
    unsigned char *buf = malloc((size_t)w * h * 3);
    if (!buf || fread(buf, 1, (size_t)w * h * 3, f) != (size_t)w * h * 3) { free(buf); fclose(f); return NULL; }
    fclose(f);
    *width = w; *height = h;
    return buf;
}
