/*
 * mort_scenes.c -- the reference's ten built-in scenes (mort.cu:129-631) and
 * the id -> scene switch (mort.cu:649-689), written against the C scene layer.
 * Scene ids keep the reference meaning:
 *   1 random_spheres   2 two_spheres   3 earth   4 two_perlin_spheres
 *   5 quads   6 cornell_box   7 cornell_smoke   8 final_scene(800,1000,40)
 *   9 final_scene(400,250,4)   10 out_of_order_spheres(35)
 */
#include "mort_host.h"
#include "mort_vec.h"

mort_vec3 mort_host_random_vec3(mort_host_rng *g);
mort_vec3 mort_host_random_vec3_range(mort_host_rng *g, float mn, float mx);

#define SOLID MORT_TEXTURE_SOLID
#define LAMB MORT_MAT_LAMBERTIAN

/* `color::random() * color::random()` (mort.cu:152,259): the two operands are
 * sibling arguments of operator*; the product is commutative, so only which
 * call draws first depends on the profile. */
static mort_vec3 random_albedo_product(mort_host_rng *g) {
    mort_vec3 first = mort_host_random_vec3(g);
    mort_vec3 second = mort_host_random_vec3(g);
    return g->args_rtl ? v_mul(second, first) : v_mul(first, second);
}

static void cam_book1_view(mort_camera *cam) { /* shared by scenes 1, 2, 4, 10 */
    cam->aspect_ratio = (float)(16.0 / 9.0);
    cam->image_width = 1200;
    cam->light_obj_type = -1;
    cam->vfov = 20;
    cam->lookfrom = v3(13, 2, 3);
    cam->lookat = v3(0, 0, 0);
    cam->vup = v3(0, 1, 0);
    cam->defocus_angle = 0;
}

static void random_spheres(mort_world *w, mort_camera *cam, mort_host_rng *g) { /* mort.cu:129-226 */
    int spheres = mort_add_hittable_list(w, true);

    int c1 = mort_add_solid_color(w, v3(.2f, .3f, .1f));
    int c2 = mort_add_solid_color(w, v3(.9f, .9f, .9f));
    int checker = mort_add_checker_texture(w, 0.32f, SOLID, c1, SOLID, c2);
    int ground_mat = mort_add_lambertian(w, MORT_TEXTURE_CHECKER, checker);
    int ground = mort_add_sphere(w, v3(0, -1000, 0), 1000, LAMB, ground_mat, true);
    mort_list_add(w, spheres, MORT_OBJ_SPHERE, ground);

    for (int a = -11; a < 11; a++) {
        for (int b = -11; b < 11; b++) {
            float choose_mat = mort_host_random_float(g);
            /* point3 center(a + 0.9*random_float(), 0.2, b + 0.9*random_float()) */
            float r0 = mort_host_random_float(g);
            float r1 = mort_host_random_float(g);
            float rx = g->args_rtl ? r1 : r0, rz = g->args_rtl ? r0 : r1;
            mort_vec3 center = v3((float)(a + 0.9 * rx), 0.2f, (float)(b + 0.9 * rz));

            if (v_len(v_sub(center, v3(4, 0.2f, 0))) > 0.9) {
                if (choose_mat < 0.8) {
                    mort_vec3 albedo = random_albedo_product(g);
                    mort_vec3 center2 = v_add(center, v3(0, mort_host_random_float_range(g, 0.0f, 0.5f), 0));
                    int col = mort_add_solid_color(w, albedo);
                    int mat = mort_add_lambertian(w, SOLID, col);
                    int s = mort_add_moving_sphere(w, center, center2, 0.2f, LAMB, mat, true);
                    mort_list_add(w, spheres, MORT_OBJ_SPHERE, s);
                } else if (choose_mat < 0.95) {
                    mort_vec3 albedo = mort_host_random_vec3_range(g, 0.5f, 1);
                    float fuzz = mort_host_random_float_range(g, 0.0f, 0.5f);
                    int mat = mort_add_metal(w, albedo, fuzz);
                    int s = mort_add_sphere(w, center, 0.2f, MORT_MAT_METAL, mat, true);
                    mort_list_add(w, spheres, MORT_OBJ_SPHERE, s);
                } else {
                    int mat = mort_add_dielectric(w, 1.5f);
                    int s = mort_add_sphere(w, center, 0.2f, MORT_MAT_DIELECTRIC, mat, true);
                    mort_list_add(w, spheres, MORT_OBJ_SPHERE, s);
                }
            }
        }
    }

    int m1 = mort_add_dielectric(w, 1.5f);
    mort_list_add(w, spheres, MORT_OBJ_SPHERE, mort_add_sphere(w, v3(0, 1, 0), 1.0f, MORT_MAT_DIELECTRIC, m1, true));

    int sc = mort_add_solid_color(w, v3(0.4f, 0.2f, 0.1f));
    int m2 = mort_add_lambertian(w, SOLID, sc);
    mort_list_add(w, spheres, MORT_OBJ_SPHERE, mort_add_sphere(w, v3(-4, 1, 0), 1.0f, LAMB, m2, true));

    int m3 = mort_add_metal(w, v3(0.7f, 0.6f, 0.5f), 0.0f);
    mort_list_add(w, spheres, MORT_OBJ_SPHERE, mort_add_sphere(w, v3(4, 1, 0), 1.0f, MORT_MAT_METAL, m3, true));

    mort_add_bvh(w, spheres, false);
    w->bvh_mode = true;

    cam_book1_view(cam);
    cam->samples_per_pixel = 100;
    cam->bounce_limit = 20;
    cam->focus_dist = 10.0f;
}

static void two_spheres(mort_world *w, mort_camera *cam) { /* mort.cu:228-253 */
    int c1 = mort_add_solid_color(w, v3(.2f, .3f, .1f));
    int c2 = mort_add_solid_color(w, v3(.9f, .9f, .9f));
    int checker = mort_add_checker_texture(w, 0.32f, SOLID, c1, SOLID, c2);
    int mat = mort_add_lambertian(w, MORT_TEXTURE_CHECKER, checker);
    mort_add_sphere(w, v3(0, -10, 0), 10, LAMB, mat, false);
    mort_add_sphere(w, v3(0, 10, 0), 10, LAMB, mat, false);

    cam_book1_view(cam);
    cam->samples_per_pixel = 20;
    cam->bounce_limit = 50;
}

static void out_of_order_spheres(mort_world *w, mort_camera *cam, int n, mort_host_rng *g) { /* mort.cu:255-290 */
    int spheres = mort_add_hittable_list(w, true);
    for (int i = 0; i < n; i++) {
        mort_vec3 albedo = random_albedo_product(g);
        float c = (float)(n - i);
        int col = mort_add_solid_color(w, albedo);
        int mat = mort_add_lambertian(w, SOLID, col);
        int s = mort_add_sphere(w, v3(c, c, c), 0.2f, LAMB, mat, true);
        mort_list_add(w, spheres, MORT_OBJ_SPHERE, s);
    }
    mort_add_bvh(w, spheres, false);

    cam_book1_view(cam);
    cam->samples_per_pixel = 1;
    cam->bounce_limit = 5;
    cam->focus_dist = 10.0f;
}

static void earth(mort_world *w, mort_camera *cam, const mort_scene_opts *o) { /* mort.cu:292-313 */
    int tex = mort_add_image_texture(w, o ? o->earth_texels : NULL, o ? o->earth_width : 0, o ? o->earth_height : 0);
    int mat = mort_add_lambertian(w, MORT_TEXTURE_IMAGE, tex);
    mort_add_sphere(w, v3(0, 0, 0), 2, LAMB, mat, false);

    cam->aspect_ratio = (float)(16.0 / 9.0);
    cam->image_width = 1200;
    cam->samples_per_pixel = 100;
    cam->bounce_limit = 50;
    cam->light_obj_type = -1;
    cam->vfov = 20;
    cam->lookfrom = v3(0, 0, 12);
    cam->lookat = v3(0, 0, 0);
    cam->vup = v3(0, 1, 0);
    cam->defocus_angle = 0;
}

static void two_perlin_spheres(mort_world *w, mort_camera *cam, mort_host_rng *g) { /* mort.cu:315-338 */
    int tex = mort_add_noise_texture(w, 4.0f, g);
    int mat = mort_add_lambertian(w, MORT_TEXTURE_NOISE, tex);
    mort_add_sphere(w, v3(0, -1000, 0), 1000, LAMB, mat, false);
    mort_add_sphere(w, v3(0, 2, 0), 2, LAMB, mat, false);

    cam_book1_view(cam);
    cam->samples_per_pixel = 5;
    cam->bounce_limit = 10;
}

static void quads(mort_world *w, mort_camera *cam) { /* mort.cu:340-390 */
    int red = mort_add_solid_color(w, v3(1.0f, 0.2f, 0.2f));
    int green = mort_add_solid_color(w, v3(0.2f, 1.0f, 0.2f));
    int blue = mort_add_solid_color(w, v3(0.2f, 0.2f, 1.0f));
    int orange = mort_add_solid_color(w, v3(1.0f, 0.5f, 0.0f));
    int teal = mort_add_solid_color(w, v3(0.2f, 0.8f, 0.8f));

    int left = mort_add_lambertian(w, SOLID, red);
    int back = mort_add_lambertian(w, SOLID, green);
    int right = mort_add_lambertian(w, SOLID, blue);
    int upper = mort_add_lambertian(w, SOLID, orange);
    int lower = mort_add_lambertian(w, SOLID, teal);

    mort_add_quad(w, v3(-3, -2, 5), v3(0, 0, -4), v3(0, 4, 0), LAMB, left, false);
    mort_add_quad(w, v3(-2, -2, 0), v3(4, 0, 0), v3(0, 4, 0), LAMB, back, false);
    mort_add_quad(w, v3(3, -2, 1), v3(0, 0, 4), v3(0, 4, 0), LAMB, right, false);
    mort_add_quad(w, v3(-2, 3, 1), v3(4, 0, 0), v3(0, 0, 4), LAMB, upper, false);
    mort_add_quad(w, v3(-2, -3, 5), v3(4, 0, 0), v3(0, 0, -4), LAMB, lower, false);

    cam->aspect_ratio = 1.0f;
    cam->image_width = 400;
    cam->samples_per_pixel = 100;
    cam->bounce_limit = 50;
    cam->light_obj_type = -1;
    cam->vfov = 20;
    cam->lookfrom = v3(0, 0, 9);
    cam->lookat = v3(0, 0, 0);
    cam->vup = v3(0, 1, 0);
    cam->defocus_angle = 0;
}

static void cornell_view(mort_camera *cam) {
    cam->aspect_ratio = 1.0f;
    cam->bounce_limit = 50;
    cam->background = v3(0, 0, 0);
    cam->vfov = 40;
    cam->lookfrom = v3(278, 278, -800);
    cam->lookat = v3(278, 278, 0);
    cam->vup = v3(0, 1, 0);
    cam->defocus_angle = 0;
}

static void cornell_box(mort_world *w, mort_camera *cam) { /* mort.cu:392-448 */
    /* construction order fixes the indices: textures red, white, green, light;
     * materials red_wall, white_wall, green_wall (lambertian 0..2), lamp, glass */
    int red = mort_add_solid_color(w, v3(.65f, .05f, .05f));
    int white = mort_add_solid_color(w, v3(.73f, .73f, .73f));
    int green = mort_add_solid_color(w, v3(.12f, .45f, .15f));
    int light = mort_add_solid_color(w, v3(15, 15, 10));

    int red_wall = mort_add_lambertian(w, SOLID, red);
    int white_wall = mort_add_lambertian(w, SOLID, white);
    int green_wall = mort_add_lambertian(w, SOLID, green);
    int lamp = mort_add_diffuse_light(w, SOLID, light);
    int glass = mort_add_dielectric(w, 1.5f);

    int lights = mort_add_hittable_list(w, false);
    int ceiling_lamp = mort_add_quad(w, v3(343, 554, 332), v3(-130, 0, 0), v3(0, 0, -105), MORT_MAT_DIFFUSE_LIGHT, lamp, true);
    mort_list_add(w, lights, MORT_OBJ_QUAD, ceiling_lamp);
    int glass_sphere = mort_add_sphere(w, v3(190, 90, 190), 90, MORT_MAT_DIELECTRIC, glass, true);
    mort_list_add(w, lights, MORT_OBJ_SPHERE, glass_sphere);

    mort_add_quad(w, v3(555, 0, 0), v3(0, 555, 0), v3(0, 0, 555), LAMB, green_wall, false);
    mort_add_quad(w, v3(0, 0, 0), v3(0, 555, 0), v3(0, 0, 555), LAMB, red_wall, false);
    mort_add_quad(w, v3(0, 0, 0), v3(555, 0, 0), v3(0, 0, 555), LAMB, white_wall, false);
    mort_add_quad(w, v3(555, 555, 555), v3(-555, 0, 0), v3(0, 0, -555), LAMB, white_wall, false);
    mort_add_quad(w, v3(0, 0, 555), v3(555, 0, 0), v3(0, 555, 0), LAMB, white_wall, false);

    mort_rotated_box(w, v3(165, 330, 165), v3(265, 0, 295), 15, LAMB, white_wall);

    cornell_view(cam);
    cam->image_width = 600;
    cam->samples_per_pixel = 1000;
    cam->light_obj_type = MORT_OBJ_HITTABLE_LIST;
    cam->light_obj_idx = lights;
}

static void cornell_smoke(mort_world *w, mort_camera *cam) { /* mort.cu:450-504 */
    int red = mort_add_solid_color(w, v3(.65f, .05f, .05f));
    int white = mort_add_solid_color(w, v3(.73f, .73f, .73f));
    int green = mort_add_solid_color(w, v3(.12f, .45f, .15f));
    int light = mort_add_solid_color(w, v3(15, 15, 10));
    int black_smoke_color = mort_add_solid_color(w, v3(0, 0, 0));
    int white_smoke_color = mort_add_solid_color(w, v3(1, 1, 1));

    int red_wall = mort_add_lambertian(w, SOLID, red);
    int white_wall = mort_add_lambertian(w, SOLID, white);
    int green_wall = mort_add_lambertian(w, SOLID, green);
    int lamp = mort_add_diffuse_light(w, SOLID, light);
    int black_smoke = mort_add_lambertian(w, SOLID, black_smoke_color);
    int white_smoke = mort_add_lambertian(w, SOLID, white_smoke_color);

    mort_add_quad(w, v3(555, 0, 0), v3(0, 555, 0), v3(0, 0, 555), LAMB, green_wall, false);
    mort_add_quad(w, v3(0, 0, 0), v3(0, 555, 0), v3(0, 0, 555), LAMB, red_wall, false);
    mort_add_quad(w, v3(343, 554, 332), v3(-130, 0, 0), v3(0, 0, -105), MORT_MAT_DIFFUSE_LIGHT, lamp, false);
    mort_add_quad(w, v3(0, 0, 0), v3(555, 0, 0), v3(0, 0, 555), LAMB, white_wall, false);
    mort_add_quad(w, v3(555, 555, 555), v3(-555, 0, 0), v3(0, 0, -555), LAMB, white_wall, false);
    mort_add_quad(w, v3(0, 0, 555), v3(555, 0, 0), v3(0, 555, 0), LAMB, white_wall, false);

    mort_rotated_smoke_box(w, v3(165, 330, 165), v3(265, 0, 295), 15, 0.01f, LAMB, black_smoke);
    mort_rotated_smoke_box(w, v3(165, 165, 165), v3(130, 0, 65), -18, 0.01f, LAMB, white_smoke);

    cornell_view(cam);
    cam->image_width = 800;
    cam->samples_per_pixel = 2000;
    /* mort.cu:495-496 passes the lamp *material's* tag and index as the light
     * object: type 4 reads as OBJ_ROTATE_Y. Preserved. */
    cam->light_obj_type = MORT_MAT_DIFFUSE_LIGHT;
    cam->light_obj_idx = lamp;
}

static void final_scene(mort_world *w, mort_camera *cam, int image_width, int spp, int max_depth,
                        mort_host_rng *g, const mort_scene_opts *o) { /* mort.cu:506-631 */
    int ground_color = mort_add_solid_color(w, v3(0.48f, 0.83f, 0.53f));
    int ground_mat = mort_add_lambertian(w, SOLID, ground_color);

    int boxes_per_side = 20;
    for (int i = 0; i < boxes_per_side; i++) {
        for (int j = 0; j < boxes_per_side; j++) {
            double ww = 100.0;
            double x0 = -1000.0 + i * ww;
            double z0 = -1000.0 + j * ww;
            double y0 = 0.0;
            double x1 = x0 + ww;
            float y1 = mort_host_random_float_range(g, 1, 101);
            double z1 = z0 + ww;
            mort_box(w, v3((float)x0, (float)y0, (float)z0), v3((float)x1, y1, (float)z1), LAMB, ground_mat);
        }
    }

    int light_color = mort_add_solid_color(w, v3(7, 7, 7));
    int light_mat = mort_add_diffuse_light(w, SOLID, light_color);
    int light = mort_add_quad(w, v3(123, 554, 147), v3(300, 0, 0), v3(0, 0, 265), MORT_MAT_DIFFUSE_LIGHT, light_mat, false);

    mort_vec3 center1 = v3(400, 400, 200);
    mort_vec3 center2 = v_add(center1, v3(30, 0, 0));
    int moving_color = mort_add_solid_color(w, v3(0.7f, 0.3f, 0.1f));
    int moving_mat = mort_add_lambertian(w, SOLID, moving_color);
    mort_add_moving_sphere(w, center1, center2, 50, LAMB, moving_mat, false);

    int glass_mat = mort_add_dielectric(w, 1.5f);
    mort_add_sphere(w, v3(260, 150, 45), 50, MORT_MAT_DIELECTRIC, glass_mat, false);

    int metal_mat = mort_add_metal(w, v3(0.8f, 0.8f, 0.9f), 1.0f);
    mort_add_sphere(w, v3(0, 150, 145), 50, MORT_MAT_METAL, metal_mat, false);

    int sub_color = mort_add_solid_color(w, v3(0.2f, 0.4f, 0.9f));
    int sub_mat = mort_add_lambertian(w, SOLID, sub_color);
    int sub_sphere = mort_add_sphere(w, v3(360, 150, 145), 70, MORT_MAT_DIELECTRIC, glass_mat, false);
    mort_add_constant_medium(w, MORT_OBJ_SPHERE, sub_sphere, 0.2f, LAMB, sub_mat, false);

    int boundary_color = mort_add_solid_color(w, v3(1, 1, 1));
    int boundary_mat = mort_add_lambertian(w, SOLID, boundary_color);
    int boundary_sphere = mort_add_sphere(w, v3(0, 0, 0), 5000, MORT_MAT_DIELECTRIC, glass_mat, false);
    mort_add_constant_medium(w, MORT_OBJ_SPHERE, boundary_sphere, 0.0001f, LAMB, boundary_mat, false);

    int earth_tex = mort_add_image_texture(w, o ? o->earth_texels : NULL, o ? o->earth_width : 0, o ? o->earth_height : 0);
    int earth_mat = mort_add_lambertian(w, MORT_TEXTURE_IMAGE, earth_tex);
    mort_add_sphere(w, v3(400, 200, 400), 100, LAMB, earth_mat, false);

    int noise_tex = mort_add_noise_texture(w, 0.1f, g);
    int noise_mat = mort_add_lambertian(w, MORT_TEXTURE_NOISE, noise_tex);
    mort_add_sphere(w, v3(220, 280, 300), 80, LAMB, noise_mat, false);

    /* the reference constructs cluster_color/cluster_mat before the loop and
     * adds them after it; indices are fixed at construction */
    int cluster_color = mort_add_solid_color(w, v3(.73f, .73f, .73f));
    int cluster_mat = mort_add_lambertian(w, SOLID, cluster_color);
    int ns = 1000;
    int cluster_base = mort_add_hittable_list(w, true);
    for (int j = 0; j < ns; j++) {
        int s = mort_add_sphere(w, mort_host_random_vec3_range(g, 0, 165), 10, LAMB, cluster_mat, true);
        mort_list_add(w, cluster_base, MORT_OBJ_SPHERE, s);
    }
    int cluster_rotate = mort_add_rotate_y(w, MORT_OBJ_HITTABLE_LIST, cluster_base, 15, true);
    mort_add_translate(w, MORT_OBJ_ROTATE_Y, cluster_rotate, v3(-100, 270, 395), false);

    cam->aspect_ratio = 1.0f;
    cam->image_width = image_width;
    cam->samples_per_pixel = spp;
    cam->bounce_limit = max_depth;
    cam->background = v3(0, 0, 0);
    cam->light_obj_type = MORT_OBJ_QUAD;
    cam->light_obj_idx = light;
    cam->vfov = 40;
    cam->lookfrom = v3(478, 278, -600);
    cam->lookat = v3(278, 278, 0);
    cam->vup = v3(0, 1, 0);
    cam->defocus_angle = 0;
}

int mort_scene_build(int id, mort_world *w, mort_camera *cam, const mort_scene_opts *o) {
    mort_host_rng g;
    mort_host_rng_init(&g, 1u, o ? o->args_rtl : 0); /* the reference never calls srand() */
    mort_camera_defaults(cam);
    switch (id) { /* mort.cu:649-689 */
    case 1: random_spheres(w, cam, &g); break;
    case 2: two_spheres(w, cam); break;
    case 3: earth(w, cam, o); break;
    case 4: two_perlin_spheres(w, cam, &g); break;
    case 5: quads(w, cam); break;
    case 6: cornell_box(w, cam); break;
    case 7: cornell_smoke(w, cam); break;
    case 8: final_scene(w, cam, 800, 1000, 40, &g, o); break;
    case 9: final_scene(w, cam, 400, 250, 4, &g, o); break;
    case 10: out_of_order_spheres(w, cam, 35, &g); break;
    default: break; /* no default: in the reference: empty world */
    }
    return 0;
}
