/*
 * mort -- command line of the renderer: `mort <scene_id>` as in the reference
 * (mort.cu:633-689), plus overrides and file output (the reference only draws
 * into a GLUT window).  Host code is C; the render goes through the C ABI of
 * libmort_hip.so.  Errors print and exit like HANDLE_ERROR (include/book.h:21-30).
 *
 *   mort <scene_id> [--width W] [--aspect A] [--spp N] [--depth D] [--seed S]
 *                   [--out file.ppm] [--dump-f32 file.raw] [--states-in f] [--states-out f]
 *                   [--earth file.ppm] [--rtl] [--frames N] [--device K] [--mode mega|wave]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mort_hip.h"
#include "mort_host.h"

static void die(mort_ctx *ctx, int st, const char *what) {
    fprintf(stderr, "%s: %s %s\n", what, mort_hip_strerror(st), ctx ? mort_hip_last_error(ctx) : "");
    exit(EXIT_FAILURE);
}

int main(int argc, char **argv) {
    if (argc < 2) {
        printf("Usage: mort <number_between_1_and_10> [--width W] [--aspect A] [--spp N] [--depth D] [--seed S] "
               "[--out f.ppm] [--dump-f32 f.raw] [--states-in f] [--states-out f] [--earth f.ppm] [--rtl] [--frames N]\n");
        return -1;
    }
    int scene = atoi(argv[1]);
    int width = 0, spp = 0, depth = -1, frames = 1, device = 0, rtl = 0, mode = MORT_MODE_MEGA;
    double aspect = 0;
    unsigned long long seed = MORT_DEFAULT_SEED;
    const char *out = NULL, *dump = NULL, *sin = NULL, *sout = NULL, *earth = "tests/golden/earthmap.ppm";
    for (int i = 2; i < argc; i++) {
#define ARG(name) (strcmp(argv[i], name) == 0 && i + 1 < argc)
        if (ARG("--width")) width = atoi(argv[++i]);
        else if (ARG("--aspect")) aspect = atof(argv[++i]);
        else if (ARG("--spp")) spp = atoi(argv[++i]);
        else if (ARG("--depth")) depth = atoi(argv[++i]);
        else if (ARG("--seed")) seed = strtoull(argv[++i], NULL, 10);
        else if (ARG("--out")) out = argv[++i];
        else if (ARG("--dump-f32")) dump = argv[++i];
        else if (ARG("--states-in")) sin = argv[++i];
        else if (ARG("--states-out")) sout = argv[++i];
        else if (ARG("--earth")) earth = argv[++i];
        else if (ARG("--frames")) frames = atoi(argv[++i]);
        else if (ARG("--device")) device = atoi(argv[++i]);
        else if (ARG("--mode")) { const char *m = argv[++i]; mode = (strcmp(m, "wave") == 0) ? MORT_MODE_WAVE : MORT_MODE_MEGA; }
        else if (strcmp(argv[i], "--rtl") == 0) rtl = 1;
        else { fprintf(stderr, "unknown option %s\n", argv[i]); return -1; }
    }

    mort_world world;
    mort_camera cam;
    if (mort_world_init(&world) != 0) { fprintf(stderr, "out of memory\n"); return EXIT_FAILURE; }
    mort_scene_opts opts;
    memset(&opts, 0, sizeof opts);
    opts.args_rtl = rtl;
    unsigned char *texels = NULL;
    if (scene == 3 || scene == 8 || scene == 9) {
        texels = mort_read_ppm(earth, &opts.earth_width, &opts.earth_height);
        if (!texels) fprintf(stderr, "ERROR: Could not load image file '%s'.\n", earth); /* img_loader.h:33 */
        opts.earth_texels = texels;
    }
    mort_scene_build(scene, &world, &cam, &opts);
    if (width > 0) cam.image_width = width;
    if (aspect > 0) cam.aspect_ratio = (float)aspect;
    if (spp > 0) cam.samples_per_pixel = spp;
    if (depth >= 0) cam.bounce_limit = depth;
    mort_camera_initialize(&cam);
    const int W = cam.image_width, H = cam.image_height;

    mort_ctx *ctx = NULL;
    int st = mort_hip_init(device, &ctx);
    if (st != MORT_OK) die(NULL, st, "mort_hip_init");
    if ((st = mort_hip_upload_world(ctx, &world)) != MORT_OK) die(ctx, st, "mort_hip_upload_world");

    size_t npx = (size_t)W * H;
    if (sin) {
        mort_rng_state *s = malloc(npx * sizeof *s);
        FILE *f = fopen(sin, "rb");
        if (!s || !f || fread(s, sizeof *s, npx, f) != npx) { fprintf(stderr, "cannot read %zu states from %s\n", npx, sin); return EXIT_FAILURE; }
        fclose(f);
        if ((st = mort_hip_rng_load(ctx, s, W, H)) != MORT_OK) die(ctx, st, "mort_hip_rng_load");
        free(s);
    } else if ((st = mort_hip_rng_seed(ctx, seed, W, H)) != MORT_OK) die(ctx, st, "mort_hip_rng_seed");

    uint8_t *rgba = calloc(npx, 4);
    float *accum = dump ? calloc(npx * 3, sizeof(float)) : NULL;
    double total_ms = 0;
    mort_stats stats;
    for (int f = 0; f < frames; f++) {
        if ((st = mort_hip_render(ctx, &cam, mode, rgba, accum, NULL, &stats)) != MORT_OK) die(ctx, st, "mort_hip_render");
        total_ms += stats.seconds * 1e3;
        printf("Avg. time per frame: %3.1f ms\n", total_ms / (f + 1)); /* mort.cu:119 */
    }
    int eff = mort_camera_effective_spp(&cam);
    printf("{\"scene\": %d, \"width\": %d, \"height\": %d, \"spp_nominal\": %d, \"spp_effective\": %d, \"depth\": %d, "
           "\"seconds\": %.6f, \"msamples_per_s\": %.3f, \"segments\": %llu, \"segments_per_s\": %.4g, "
           "\"algorithmic_hbm_bytes\": %llu, \"hbm_GBps\": %.4g, \"hbm_frac_of_8TBps\": %.3g, \"reference_walks\": %llu}\n",
           scene, W, H, cam.samples_per_pixel, eff, cam.bounce_limit, stats.seconds,
           (double)npx * eff / stats.seconds / 1e6, (unsigned long long)stats.segments, stats.segments / stats.seconds,
           (unsigned long long)stats.algorithmic_hbm_bytes, stats.algorithmic_hbm_bytes / stats.seconds / 1e9,
           stats.algorithmic_hbm_bytes / stats.seconds / 8e12, (unsigned long long)stats.reference_walks);
    if (out && mort_write_ppm(out, rgba, W, H) != 0) { fprintf(stderr, "cannot write %s\n", out); return EXIT_FAILURE; }
    if (dump) {
        FILE *f = fopen(dump, "wb");
        if (!f || fwrite(accum, sizeof(float), npx * 3, f) != npx * 3) { fprintf(stderr, "cannot write %s\n", dump); return EXIT_FAILURE; }
        fclose(f);
    }
    if (sout) {
        mort_rng_state *s = malloc(npx * sizeof *s);
        if ((st = mort_hip_rng_store(ctx, s, W, H)) != MORT_OK) die(ctx, st, "mort_hip_rng_store");
        FILE *f = fopen(sout, "wb");
        if (!f || fwrite(s, sizeof *s, npx, f) != npx) { fprintf(stderr, "cannot write %s\n", sout); return EXIT_FAILURE; }
        fclose(f);
        free(s);
    }
    mort_hip_shutdown(ctx);
    mort_world_free(&world);
    free(texels); free(rgba); free(accum);
    return 0;
}
