/*
 * mort_host.h -- host-side scene layer (C): world containers, object /
 * material / texture constructors, host BVH builder, camera set-up, the ten
 * built-in scenes and the host random source they draw from.
 *
 * This is the counterpart of the reference's host code above the kernel
 * launch: world::add overloads (world.cuh:27-90), object constructors
 * (objects.cuh:38-55,170-185,258-263,296-329,384-394,459-469), the BVH
 * builder (objects.cuh:529-661,981-1006), box helpers (utils.h:51-126),
 * Camera::initialize (camera.cuh:47-84) and the scene catalogue
 * (mort.cu:129-631,649-689).  Built as libmort_host.so; it has no GPU
 * dependency and is used by the `mort` CLI, the tests and the CPU oracle.
 *
 * The reference gives every object an index from a per-type static counter at
 * construction and later copies it into the world at position num_X++; all
 * built-in scenes construct and add in the same order, so here construct+add
 * is one call that returns that index.
 */
#ifndef MORT_HOST_H
#define MORT_HOST_H

#include "mort_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- host random source: model of the MSVC CRT rand() the reference's
 * random_float()/random_int() sit on (rng.cuh:44-58), never seeded => h0 = 1.
 * args_rtl selects the order in which a compiler evaluates the random_float()
 * calls that appear as sibling function arguments (mort.cu:147,152,164;
 * vec3.cuh:63-69; textures.cuh:166): 0 = left-to-right, 1 = right-to-left. */
typedef struct mort_host_rng {
    uint32_t state;
    int args_rtl;
} mort_host_rng;

void mort_host_rng_init(mort_host_rng *g, uint32_t seed, int args_rtl);
int mort_host_rand(mort_host_rng *g);                     /* 0..32767 */
float mort_host_random_float(mort_host_rng *g);           /* rand()/(RAND_MAX+1.0) */
float mort_host_random_float_range(mort_host_rng *g, float mn, float mx);

/* ---- world lifecycle (world.cuh:20-25,92-96) ---- */
int mort_world_init(mort_world *w);   /* allocates every host array at capacity */
void mort_world_free(mort_world *w);

/* ---- textures / materials: return the new element's idx, or -1 when full ---- */
int mort_add_solid_color(mort_world *w, mort_vec3 c);
int mort_add_checker_texture(mort_world *w, float scale, int even_type, int even_idx,
                             int odd_type, int odd_idx);
int mort_add_image_texture(mort_world *w, const unsigned char *texels, int width, int height);
int mort_add_noise_texture(mort_world *w, float scale, mort_host_rng *g);
int mort_add_lambertian(mort_world *w, int tex_type, int tex_idx);
int mort_add_metal(mort_world *w, mort_vec3 albedo, float fuzz);
int mort_add_dielectric(mort_world *w, float refraction_index);
int mort_add_diffuse_light(mort_world *w, int tex_type, int tex_idx);
int mort_add_isotropic(mort_world *w, int tex_type, int tex_idx);

/* ---- objects ---- */
int mort_add_sphere(mort_world *w, mort_vec3 center, float radius, int mat_type, int mat_idx, bool skip);
int mort_add_moving_sphere(mort_world *w, mort_vec3 cen1, mort_vec3 cen2, float radius,
                           int mat_type, int mat_idx, bool skip);
int mort_add_quad(mort_world *w, mort_vec3 Q, mort_vec3 u, mort_vec3 v, int mat_type, int mat_idx, bool skip);
int mort_add_translate(mort_world *w, int obj_type, int obj_idx, mort_vec3 displacement, bool skip);
int mort_add_rotate_y(mort_world *w, int obj_type, int obj_idx, float theta_degrees, bool skip);
int mort_add_constant_medium(mort_world *w, int obj_type, int obj_idx, float density,
                             int mat_type, int mat_idx, bool skip);
int mort_add_hittable_list(mort_world *w, bool skip);
int mort_list_add(mort_world *w, int list_idx, int obj_type, int obj_idx);
/* Builds a BVH over list `list_idx` (median split on the largest axis; same-type
 * primitives are physically reordered in the world arrays) and adds it, which
 * also sets bvh_mode (world.cuh:51-54). */
int mort_add_bvh(mort_world *w, int list_idx, bool skip);

mort_aabb mort_get_bbox(const mort_world *w, int obj_type, int obj_idx);

/* ---- box helpers (utils.h:51-126) ---- */
void mort_box(mort_world *w, mort_vec3 a, mort_vec3 b, int mat_type, int mat_idx);
void mort_rotated_box(mort_world *w, mort_vec3 size, mort_vec3 translation, float theta,
                      int mat_type, int mat_idx);
void mort_rotated_smoke_box(mort_world *w, mort_vec3 size, mort_vec3 translation, float theta,
                            float density, int mat_type, int mat_idx);

/* ---- camera ---- */
void mort_camera_defaults(mort_camera *cam);    /* in-class initialisers, camera.cuh:13-43 */
void mort_camera_initialize(mort_camera *cam);  /* camera.cuh:47-84 */
/* One idle tick of the reference's input() (mort.cu:49-91: W/S/A/D move lookfrom and lookat one unit along the camera basis,
 * dragging with the left button turns lookat by -delta / 500 rad about vup / u), then initialize(): scripted, headless. */
#define MORT_KEY_W 1
#define MORT_KEY_S 2
#define MORT_KEY_A 4
#define MORT_KEY_D 8
void mort_camera_input(mort_camera *cam, int keys, int mouse_dx, int mouse_dy, int left_button);
/* effective samples per pixel: floor(sqrt(spp))^2 (camera.cuh:51,187-188) */
int mort_camera_effective_spp(const mort_camera *cam);

/* ---- scene catalogue (mort.cu:649-689) ---- */
typedef struct mort_scene_opts {
    int args_rtl;                      /* host RNG argument-order profile */
    const unsigned char *earth_texels; /* decoded imgs/earthmap.jpg (scenes 3, 8, 9) or NULL */
    int earth_width, earth_height;
} mort_scene_opts;

/* Fills `w` (already mort_world_init'ed) and `cam` (parameters only; call
 * mort_camera_initialize afterwards) for scene 1..10.  Any other id leaves the
 * world empty, as the reference's switch has no default.  Returns 0. */
int mort_scene_build(int scene_id, mort_world *w, mort_camera *cam, const mort_scene_opts *opts);

/* ---- image file helpers ---- */
/* Binary PPM (P6) writer; the framebuffer's row 0 is the bottom image row
 * (SURVEY A.5), so rows are emitted H-1..0. rgba = W*H*4 bytes. */
int mort_write_ppm(const char *path, const uint8_t *rgba, int width, int height);
/* Reads a P6 PPM into a malloc'ed tightly packed RGB buffer (caller frees). */
unsigned char *mort_read_ppm(const char *path, int *width, int *height);
/* Baseline JPEG -> malloc'ed tightly packed RGB (caller frees), NULL on failure: the image_texture input of scenes 3, 8, 9
 * (img_loader.h:38-44 loads imgs/earthmap.jpg through stb_image; same integer pipeline, same bytes -- mort_jpeg.c). */
unsigned char *mort_decode_jpeg(const unsigned char *data, size_t size, int *width, int *height);
unsigned char *mort_read_jpeg(const char *path, int *width, int *height);
/* By extension: .jpg / .jpeg through mort_read_jpeg, anything else through mort_read_ppm. */
unsigned char *mort_read_image(const char *path, int *width, int *height);

#ifdef __cplusplus
}
#endif
#endif /* MORT_HOST_H */
