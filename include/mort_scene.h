/*
 * mort_scene.h -- plain-data scene, camera, material and texture structs.
 *
 * These are the structs the reference's host code fills in and hands to its
 * render kernel by value / by cudaMemcpyToSymbol.  Field names, order, tags
 * and capacities follow the reference member lists so that the host side of
 * the reference can populate them unchanged; sizes are static-asserted
 * against the x86-64 layout of the reference structs.  Cross references are
 * (int type, int idx) pairs, never pointers.
 *
 *   vec3 / interval / aabb        vec3.cuh:13-74, interval.cuh:6-44, aabb.cuh:7-74
 *   sphere, quad, translate, ...  objects.cuh:147-161, 237-249, 280-287,
 *                                 368-375, 440-448, 510-518, 725-735
 *   materials                     materials.cuh:28-202
 *   textures                      textures.cuh:17-266
 *   containers                    objects.cuh:767-787, materials.cuh:225-239,
 *                                 textures.cuh:286-297, world.cuh:173-178
 *   Camera                        camera.cuh:13-45
 *
 * Everything here is C (C99) and is also included from HIP/C++ sources.
 */
#ifndef MORT_SCENE_H
#define MORT_SCENE_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- object / material / texture tags (objects.cuh:13-19, materials.cuh:14-18,
 *      textures.cuh:10-13) ---- */
#define MORT_OBJ_SPHERE 1
#define MORT_OBJ_QUAD 2
#define MORT_OBJ_TRANSLATE 3
#define MORT_OBJ_ROTATE_Y 4
#define MORT_OBJ_CONSTANT_MEDIUM 5
#define MORT_OBJ_HITTABLE_LIST 6
#define MORT_OBJ_BVH 7

#define MORT_MAT_LAMBERTIAN 1
#define MORT_MAT_METAL 2
#define MORT_MAT_DIELECTRIC 3
#define MORT_MAT_DIFFUSE_LIGHT 4
#define MORT_MAT_ISOTROPIC 5

#define MORT_TEXTURE_SOLID 1
#define MORT_TEXTURE_CHECKER 2
#define MORT_TEXTURE_IMAGE 3
#define MORT_TEXTURE_NOISE 4

/* ---- capacities (objects.cuh:451,521,746-765; materials.cuh:210-223;
 *      textures.cuh:158,274-284) ---- */
#define MORT_LIST_MAX_OBJS 1000
#define MORT_MAX_BVH_NODES 1024
#define MORT_NUM_SPHERES 1100
#define MORT_NUM_QUADS 2500
#define MORT_NUM_TRANSLATE 50
#define MORT_NUM_ROTATE_Y 50
#define MORT_NUM_CONSTANT_MEDIUM 50
#define MORT_NUM_HITTABLE_LIST 2
#define MORT_NUM_BVH 2
#define MORT_NUM_LAMBERTIANS 600
#define MORT_NUM_METALS 100
#define MORT_NUM_DIELECTRICS 50
#define MORT_NUM_DIFFUSE_LIGHTS 20
#define MORT_NUM_ISOTROPICS 20
#define MORT_NUM_SOLID_COLOR 400
#define MORT_NUM_CHECKER_TEXTURE 400
#define MORT_NUM_IMAGE_TEXTURE 400
#define MORT_NUM_NOISE_TEXTURE 1
#define MORT_POINT_COUNT 256

/* Deepest bounce_limit the render entry points accept (the reference sizes its
 * per-bounce scratch by bounce_limit with no bound, mort.cu:712-725; the
 * built-in scenes use at most 50). */
#define MORT_MAX_BOUNCE_LIMIT 64

/* ---- math PODs ---- */
typedef struct mort_vec3 { float e[3]; } mort_vec3;
typedef struct mort_interval { float imin, imax; } mort_interval;
typedef struct mort_aabb { mort_interval x, y, z; } mort_aabb;

/* ---- objects ---- */
typedef struct mort_sphere {
    mort_vec3 center1;
    float radius;
    bool moves;
    mort_vec3 center_vec;
    int mat_type;
    int mat_idx;
    int idx;
    bool skip;
    mort_aabb bbox;
} mort_sphere;

typedef struct mort_quad {
    mort_vec3 Q;
    mort_vec3 u, v;
    mort_vec3 normal;
    mort_vec3 w;
    mort_aabb bbox;
    float D;
    float area;
    int mat_type;
    int mat_idx;
    int idx;
    bool skip;
} mort_quad;

typedef struct mort_translate {
    int obj_type, obj_idx;
    mort_vec3 offset;
    mort_aabb bbox;
    int idx;
    bool skip;
} mort_translate;

typedef struct mort_rotate_y {
    int obj_type, obj_idx;
    float sin_theta, cos_theta;
    mort_aabb bbox;
    int idx;
    bool skip;
} mort_rotate_y;

typedef struct mort_constant_medium {
    int obj_type, obj_idx;
    double neg_inv_density;
    int mat_type, mat_idx;
    mort_aabb bbox;
    int idx;
    bool skip;
} mort_constant_medium;

typedef struct mort_hittable_list {
    int obj_types[MORT_LIST_MAX_OBJS], obj_idxs[MORT_LIST_MAX_OBJS];
    int num_objs;
    int idx;
    bool skip;
    mort_aabb bbox;
} mort_hittable_list;

typedef struct mort_bvh {
    /* indexed per node id */
    int left_children_types[MORT_MAX_BVH_NODES], left_children_idxs[MORT_MAX_BVH_NODES];
    int right_children_types[MORT_MAX_BVH_NODES], right_children_idxs[MORT_MAX_BVH_NODES];
    bool is_internal_node[MORT_MAX_BVH_NODES];
    mort_aabb bounding_boxes[MORT_MAX_BVH_NODES];
    int idx;
    bool skip;
} mort_bvh;

/* ---- materials ---- */
typedef struct mort_lambertian { int texType, texIdx; int idx; } mort_lambertian;
typedef struct mort_metal { mort_vec3 albedo; float fuzz; int idx; } mort_metal;
typedef struct mort_dielectric { float ior, inv_ior; mort_vec3 albedo; int idx; } mort_dielectric;
typedef struct mort_diffuse_light { int texType, texIdx; int idx; } mort_diffuse_light;
typedef struct mort_isotropic { int texType, texIdx; int idx; } mort_isotropic;

/* ---- textures ---- */
typedef struct mort_solid_color { mort_vec3 color_value; int idx; } mort_solid_color;

typedef struct mort_checker_texture {
    float inv_scale;
    int evenTextureType;
    int evenTextureIdx;
    int oddTextureType;
    int oddTextureIdx;
    int idx;
} mort_checker_texture;

/* The reference keeps a cudaTextureObject_t over a pitch-2D copy of the
 * decoded 8-bit RGB image (textures.cuh:89-127,152).  Here the same 8 bytes
 * hold a pointer to the caller-owned tightly packed RGB texels
 * (width*height*3 bytes, row 0 = top scanline as decoded). */
typedef struct mort_image_texture {
    const unsigned char *texels;
    int width, height;
    int idx;
} mort_image_texture;

typedef struct mort_noise_texture {
    mort_vec3 ranvec[MORT_POINT_COUNT];
    int perm_x[MORT_POINT_COUNT];
    int perm_y[MORT_POINT_COUNT];
    int perm_z[MORT_POINT_COUNT];
    float scale;
    int idx;
} mort_noise_texture;

/* ---- containers ---- */
typedef struct mort_world_objects {
    mort_sphere *host_sphere;                   int num_spheres;
    mort_quad *host_quad;                       int num_quads;
    mort_translate *host_translate;             int num_translates;
    mort_rotate_y *host_rotate_y;               int num_rotate_y;
    mort_constant_medium *host_constant_medium; int num_constant_medium;
    mort_hittable_list *host_hittable_list;     int num_hittable_list;
    mort_bvh *host_bvh;                         int num_bvh;
} mort_world_objects;

typedef struct mort_world_materials {
    mort_lambertian *host_lambertian;       int num_lambertians;
    mort_metal *host_metal;                 int num_metals;
    mort_dielectric *host_dielectric;       int num_dielectrics;
    mort_diffuse_light *host_diffuse_light; int num_diffuse_lights;
    mort_isotropic *host_isotropic;         int num_isotropics;
} mort_world_materials;

typedef struct mort_world_textures {
    mort_solid_color *host_solid_color;         int num_solid_colors;
    mort_checker_texture *host_checker_texture; int num_checker_textures;
    mort_image_texture *host_image_texture;     int num_image_textures;
    mort_noise_texture *host_noise_texture;     int num_noise_textures;
} mort_world_textures;

typedef struct mort_world {
    mort_world_objects objs;
    mort_world_materials mats;
    mort_world_textures texs;
    bool bvh_mode;
} mort_world;

/* ---- camera (camera.cuh:12-45).  The four pointers are the reference's
 * per-bounce global scratch arrays (mort.cu:712-725); this implementation
 * keeps the bounce stack on chip and ignores them, they only keep the layout. */
typedef struct mort_camera {
    float aspect_ratio;
    int image_width;
    int image_height;
    int samples_per_pixel;
    float pixel_samples_scale;
    int sqrt_spp;
    float recip_sqrt_spp;
    int bounce_limit;
    int vfov;
    mort_vec3 background;

    void *recursionAttenuation;
    void *recursionEmission;
    void *recursionScatteringPdf;
    void *recursionPdf;

    int light_obj_type;
    int light_obj_idx;

    mort_vec3 center;
    mort_vec3 pixel00_loc;
    mort_vec3 pixel_delta_u;
    mort_vec3 pixel_delta_v;

    mort_vec3 lookfrom;
    mort_vec3 lookat;
    mort_vec3 vup;
    mort_vec3 v, u, w;

    float defocus_angle;
    float focus_dist;
    mort_vec3 defocus_disk_u;
    mort_vec3 defocus_disk_v;
} mort_camera;

/* ---- per-pixel RNG state: the 48-byte curandStateXORWOW layout the
 * reference allocates per pixel (rng.cuh:8-15, mort.cu:706-709).  Only d and
 * v[] are live on this path (no normal variates are drawn). ---- */
typedef struct mort_rng_state {
    unsigned int d;
    unsigned int v[5];
    int boxmuller_flag;
    int boxmuller_flag_double;
    float boxmuller_extra;
    double boxmuller_extra_double;
} mort_rng_state;

#define MORT_DEFAULT_SEED 69420ULL /* mort.cu:707 */

#ifdef __cplusplus
#define MORT_SA(c, m) static_assert(c, m)
#else
#define MORT_SA(c, m) _Static_assert(c, m)
#endif
MORT_SA(sizeof(mort_sphere) == 72, "sphere layout");
MORT_SA(sizeof(mort_quad) == 108, "quad layout");
MORT_SA(sizeof(mort_translate) == 52, "translate layout");
MORT_SA(sizeof(mort_rotate_y) == 48, "rotate_y layout");
MORT_SA(sizeof(mort_constant_medium) == 56, "constant_medium layout");
MORT_SA(sizeof(mort_hittable_list) == 8036, "hittable_list layout");
MORT_SA(sizeof(mort_bvh) == 41992, "bvh layout");
MORT_SA(sizeof(mort_lambertian) == 12 && sizeof(mort_metal) == 20 &&
        sizeof(mort_dielectric) == 24 && sizeof(mort_diffuse_light) == 12 &&
        sizeof(mort_isotropic) == 12, "material layout");
MORT_SA(sizeof(mort_solid_color) == 16 && sizeof(mort_checker_texture) == 24 &&
        sizeof(mort_image_texture) == 24 && sizeof(mort_noise_texture) == 6152, "texture layout");
MORT_SA(sizeof(mort_world) == 264, "world layout");
MORT_SA(sizeof(mort_camera) == 240, "camera layout");
MORT_SA(sizeof(mort_rng_state) == 48, "rng state layout");

#ifdef __cplusplus
}
#endif
#endif /* MORT_SCENE_H */
