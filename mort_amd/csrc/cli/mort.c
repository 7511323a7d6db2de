/*
 * mort -- command line of the renderer: `mort <scene_id>` as in the reference (mort.cu:633-689), plus overrides, file
 * output (the reference only draws into a GLUT window) and the multi-GPU / host modes the north_star names.  Host code
 * is C; every render goes through the C ABI of libmort_hip.so.  Errors print and exit like HANDLE_ERROR
 * (include/book.h:21-30).
 *
 *   mort <scene_id> [--width W] [--aspect A] [--spp N] [--depth D] [--seed S] [--frames N]
 *                   [--mode mega|wave|host|throughput] [--threads T] [--tree]       host: the kernel body as a host loop (no GPU)
 *                   [--gpus N] [--devices a,b,..] [--gather rccl|shm]     one process per GPU, rows partitioned, one gather
 *                   [--out f.ppm] [--dump-f32 f.raw] [--states-in f] [--states-out f] [--earth img] [--rtl] [--device K]
 *                   [--keys WASD..] [--mouse dx,dy]        the reference's interactive loop, scripted: one idle tick per frame
 *
 * --frames N re-renders like the reference's idle loop (mort.cu:93-120): RNG streams continue from frame to frame; before each
 * frame after the first, input() runs (mort.cu:49-91) with the frame's character of --keys held down ('.' = none) and the
 * --mouse delta dragged with the left button.  --out / --dump-f32 hold the last frame.
 *
 * --gpus N: N - 1 ranks are forked BEFORE any HIP call (a process that has initialised the GPU must not fork or exec);
 * rank r renders row blocks r, r + N, ... on device r (or --devices) and the packed rows are gathered to rank 0 -- over
 * RCCL (`--gather rccl`, the default: xGMI inside a node), or through a shared host mapping (`--gather shm`: for ranks
 * that share one GPU, where RCCL refuses a communicator; used by the tests on a one-GPU box).
 */
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/prctl.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include "mort_hip.h"
#include "mort_host.h"

/* ---- `--gpus N`: no rank may outlive a failed peer.  The frame gather is collective (rank 0 blocks in ncclRecv + a stream wait, the
 * peers in ncclSend), so a rank that fails takes the others down instead of leaving them waiting: a forked rank signals rank 0
 * (SIGUSR1), whose handler kills every rank and exits non-zero; rank 0 kills its ranks before it exits itself; and every forked rank
 * asks the kernel for SIGKILL when rank 0 dies for any other reason (PR_SET_PDEATHSIG). ---- */
static pid_t g_kids[64];
static int g_nkids = 0, g_rank = 0;
static void kill_ranks(void) { for (int r = 1; r <= g_nkids; r++) if (g_kids[r] > 0) kill(g_kids[r], SIGKILL); }
static void on_rank_failed(int sig) { (void)sig; kill_ranks(); static const char m[] = "a rank failed\n"; if (write(2, m, sizeof m - 1) < 0) { } _exit(EXIT_FAILURE); }
static void fail_exit(void) {
    if (g_rank == 0) kill_ranks();
    else kill(getppid(), SIGUSR1);
    exit(EXIT_FAILURE);
}
static void die(mort_ctx *ctx, int st, const char *what) {
    fprintf(stderr, "%s: %s %s\n", what, mort_hip_strerror(st), ctx ? mort_hip_last_error(ctx) : "");
    fail_exit();
}
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

static int usage(void) {
    printf("Usage: mort <number_between_1_and_10> [--width W] [--aspect A] [--spp N] [--depth D] [--seed S] [--frames N] "
           "[--mode mega|wave|host|throughput] [--threads T] [--tree] [--gpus N] [--devices a,b,..] [--gather rccl|shm] "
           "[--out f.ppm] [--dump-f32 f.raw] [--states-in f] [--states-out f] [--earth image.jpg|.ppm] [--rtl] [--device K] [--keys WASD..] [--mouse dx,dy]\n");
    return -1;
}

/* the reference loads "imgs/earthmap.jpg" relative to the working directory (mort.cu:296,599) */
static unsigned char *load_earth(const char *path, const char *argv0, int *w, int *h, char *used, size_t used_len) {
    if (path) { snprintf(used, used_len, "%s", path); return mort_read_image(path, w, h); }
    const char *cands[3] = {"imgs/earthmap.jpg", "tests/golden/earthmap.jpg", NULL};
    char rel[4096];
    for (int i = 0; i < 3; i++) {
        const char *p = cands[i];
        if (!p) { /* next to the binary: <repo>/mort_amd/bin/mort -> <repo>/tests/golden/earthmap.jpg */
            const char *slash = strrchr(argv0, '/');
            if (!slash) break;
            snprintf(rel, sizeof rel, "%.*s/../../tests/golden/earthmap.jpg", (int)(slash - argv0), argv0);
            p = rel;
        }
        unsigned char *t = mort_read_image(p, w, h);
        if (t) { snprintf(used, used_len, "%s", p); return t; }
    }
    snprintf(used, used_len, "imgs/earthmap.jpg");
    return NULL;
}

/* input() before frame f (mort.cu:49-91,94): the f-th character of --keys held down, the --mouse delta dragged */
static void frame_input(mort_camera *cam, const char *keys, int f, int mdx, int mdy) {
    int k = 0;
    if (keys && (size_t)(f - 1) < strlen(keys)) {
        const char ch = keys[f - 1];
        k = (ch == 'W' || ch == 'w') ? MORT_KEY_W : (ch == 'S' || ch == 's') ? MORT_KEY_S : (ch == 'A' || ch == 'a') ? MORT_KEY_A : (ch == 'D' || ch == 'd') ? MORT_KEY_D : 0;
    }
    if (k || mdx || mdy) mort_camera_input(cam, k, mdx, mdy, (mdx || mdy) ? 1 : 0);
}

int main(int argc, char **argv) {
    if (argc < 2) return usage();
    int scene = atoi(argv[1]);
    int width = 0, spp = 0, depth = -1, frames = 1, device = 0, rtl = 0, mode = MORT_MODE_MEGA, host_mode = 0, threads = 1, tree = 0;
    int gpus = 1, gather_shm = 0, devices[64], n_devices = 0;
    double aspect = 0;
    unsigned long long seed = MORT_DEFAULT_SEED;
    const char *out = NULL, *dump = NULL, *sin = NULL, *sout = NULL, *earth = NULL, *keys = NULL;
    int mouse_dx = 0, mouse_dy = 0;
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
        else if (ARG("--keys")) keys = argv[++i];
        else if (ARG("--mouse")) { if (sscanf(argv[++i], "%d,%d", &mouse_dx, &mouse_dy) != 2) { fprintf(stderr, "--mouse dx,dy\n"); return -1; } }
        else if (ARG("--device")) device = atoi(argv[++i]);
        else if (ARG("--threads")) threads = atoi(argv[++i]);
        else if (ARG("--gpus")) gpus = atoi(argv[++i]);
        else if (ARG("--devices")) {
            char *s = argv[++i];
            for (char *tok = strtok(s, ","); tok && n_devices < 64; tok = strtok(NULL, ",")) devices[n_devices++] = atoi(tok);
        }
        else if (ARG("--gather")) { const char *g = argv[++i]; if (strcmp(g, "shm") == 0) gather_shm = 1; else if (strcmp(g, "rccl") != 0) { fprintf(stderr, "unknown gather %s\n", g); return -1; } }
        else if (ARG("--mode")) {
            const char *m = argv[++i];
            if (strcmp(m, "wave") == 0) mode = MORT_MODE_WAVE;
            else if (strcmp(m, "throughput") == 0) mode = MORT_MODE_THROUGHPUT; /* non-parity: own streams per (pixel, stratum row) */
            else if (strcmp(m, "host") == 0) host_mode = 1;
            else if (strcmp(m, "mega") != 0) { fprintf(stderr, "unknown mode %s\n", m); return -1; }
        }
        else if (strcmp(argv[i], "--rtl") == 0) rtl = 1;
        else if (strcmp(argv[i], "--tree") == 0) tree = 1;
        else { fprintf(stderr, "unknown option %s\n", argv[i]); return usage(); }
    }
    if (gpus < 1 || gpus > 64 || frames < 1 || threads < 1) { fprintf(stderr, "bad --gpus / --frames / --threads\n"); return -1; }
    if (host_mode && gpus != 1) { fprintf(stderr, "--mode host runs on the host: --gpus does not apply\n"); return -1; }
    if (gpus > 1 && (dump || sin || sout)) { fprintf(stderr, "--dump-f32 / --states-in / --states-out are single-GPU options\n"); return -1; }
    if (n_devices && n_devices != gpus) { fprintf(stderr, "--devices needs %d entries\n", gpus); return -1; }

    mort_world world;
    mort_camera cam;
    if (mort_world_init(&world) != 0) { fprintf(stderr, "out of memory\n"); return EXIT_FAILURE; }
    mort_scene_opts opts;
    memset(&opts, 0, sizeof opts);
    opts.args_rtl = rtl;
    unsigned char *texels = NULL;
    if (scene == 3 || scene == 8 || scene == 9) {
        char used[4096];
        texels = load_earth(earth, argv[0], &opts.earth_width, &opts.earth_height, used, sizeof used);
        if (!texels) { /* img_loader.h:33 prints this and renders the missing-texture colour; a batch render of the wrong picture helps nobody */
            fprintf(stderr, "ERROR: Could not load image file '%s'.\n", used);
            return EXIT_FAILURE;
        }
        opts.earth_texels = texels;
    }
    mort_scene_build(scene, &world, &cam, &opts);
    if (width > 0) cam.image_width = width;
    if (aspect > 0) cam.aspect_ratio = (float)aspect;
    if (spp > 0) cam.samples_per_pixel = spp;
    if (depth >= 0) cam.bounce_limit = depth;
    mort_camera_initialize(&cam);
    const int W = cam.image_width, H = cam.image_height;
    const size_t npx = (size_t)W * H;
    const int eff = mort_camera_effective_spp(&cam);

    /* ---- ranks: fork before anything touches the GPU ---- */
    int rank = 0;
    int id_pipe[64][2];
    pid_t kids[64];
    uint8_t *shm = NULL; /* --gather shm: rank r writes its rows of the full frame here */
    if (gpus > 1) {
        if (gather_shm) {
            shm = mmap(NULL, npx * 4, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
            if (shm == MAP_FAILED) { perror("mmap"); return EXIT_FAILURE; }
        }
        for (int r = 1; r < gpus; r++) if (pipe(id_pipe[r]) != 0) { perror("pipe"); return EXIT_FAILURE; }
        signal(SIGUSR1, on_rank_failed);
        const pid_t parent = getpid();
        for (int r = 1; r < gpus; r++) {
            pid_t p = fork();
            if (p < 0) { perror("fork"); kill_ranks(); return EXIT_FAILURE; }
            if (p == 0) {
                rank = g_rank = r; g_nkids = 0;
                signal(SIGUSR1, SIG_DFL);
                prctl(PR_SET_PDEATHSIG, SIGKILL);
                if (getppid() != parent) _exit(EXIT_FAILURE); /* rank 0 died before the request took effect */
                break;
            }
            kids[r] = g_kids[r] = p; g_nkids = r;
        }
        device = n_devices ? devices[rank] : rank;
        /* test hooks for the failure path (tests/test_cli.py; no GPU needed): MORT_TEST_FAIL_RANK=r makes rank r fail here,
         * MORT_TEST_BLOCK_RANK0=1 makes rank 0 wait as it would inside the gather */
        const char *tf = getenv("MORT_TEST_FAIL_RANK");
        if (tf && atoi(tf) == rank) { if (rank > 0) usleep(200000); fprintf(stderr, "rank %d: MORT_TEST_FAIL_RANK\n", rank); fail_exit(); }
        if (tf && rank == 0 && getenv("MORT_TEST_BLOCK_RANK0")) { sleep(60); fprintf(stderr, "rank 0 was not taken down by the failed rank\n"); kill_ranks(); return 3; }
        if (tf && rank != 0) { sleep(60); _exit(3); } /* a healthy peer that would wait in ncclSend: must be killed, not waited for */
    }

    uint8_t *rgba = calloc(npx, 4);
    float *accum = dump ? calloc(npx * 3, sizeof(float)) : NULL;
    if (!rgba || (dump && !accum)) { fprintf(stderr, "out of memory\n"); fail_exit(); }
    mort_stats stats;
    memset(&stats, 0, sizeof stats);
    double total_ms = 0, frame_wall = 0;
    mort_ctx *ctx = NULL;
    mort_rng_state *hstates = NULL;
    int st;

    if (host_mode) { /* ---- the kernel body as a host loop: no GPU ---- */
        hstates = malloc(npx * sizeof *hstates);
        if (!hstates) { fprintf(stderr, "out of memory\n"); return EXIT_FAILURE; }
        if (sin) {
            FILE *f = fopen(sin, "rb");
            if (!f || fread(hstates, sizeof *hstates, npx, f) != npx) { fprintf(stderr, "cannot read %zu states from %s\n", npx, sin); return EXIT_FAILURE; }
            fclose(f);
        } else if ((st = mort_hip_rng_seed_host(seed, W, H, hstates)) != MORT_OK) die(NULL, st, "mort_hip_rng_seed_host");
        for (int f = 0; f < frames; f++) {
            if (f > 0) frame_input(&cam, keys, f, mouse_dx, mouse_dy);
            if ((st = mort_hip_render_host(&world, &cam, hstates, threads, tree ? MORT_HOST_TREE : 0, rgba, accum, NULL, &stats)) != MORT_OK) die(NULL, st, "mort_hip_render_host");
            total_ms += stats.seconds * 1e3;
            printf("Avg. time per frame: %3.1f ms\n", total_ms / (f + 1)); /* mort.cu:119 */
        }
        frame_wall = stats.seconds;
    } else {
        st = mort_hip_init(device, &ctx);
        if (st != MORT_OK) die(NULL, st, "mort_hip_init");
        if (gpus > 1) {
            mort_partition part = {rank, gpus, 8};
            if ((st = mort_hip_set_partition(ctx, &part)) != MORT_OK) die(ctx, st, "mort_hip_set_partition");
            if (!gather_shm) { /* rank 0 makes the RCCL id, the others read it from their pipe */
                unsigned char id[MORT_COMM_ID_BYTES];
                if (rank == 0) {
                    if ((st = mort_hip_comm_id(id)) != MORT_OK) die(ctx, st, "mort_hip_comm_id");
                    for (int r = 1; r < gpus; r++) if (write(id_pipe[r][1], id, sizeof id) != (ssize_t)sizeof id) { perror("write"); fail_exit(); }
                } else if (read(id_pipe[rank][0], id, sizeof id) != (ssize_t)sizeof id) { perror("read"); fail_exit(); }
                if ((st = mort_hip_comm_init(ctx, id, rank, gpus)) != MORT_OK) die(ctx, st, "mort_hip_comm_init");
            }
        }
        if ((st = mort_hip_upload_world(ctx, &world)) != MORT_OK) die(ctx, st, "mort_hip_upload_world");
        if (sin) {
            mort_rng_state *s = malloc(npx * sizeof *s);
            FILE *f = fopen(sin, "rb");
            if (!s || !f || fread(s, sizeof *s, npx, f) != npx) { fprintf(stderr, "cannot read %zu states from %s\n", npx, sin); fail_exit(); }
            fclose(f);
            if ((st = mort_hip_rng_load(ctx, s, W, H)) != MORT_OK) die(ctx, st, "mort_hip_rng_load");
            free(s);
        } else if ((st = mort_hip_rng_seed(ctx, seed, W, H)) != MORT_OK) die(ctx, st, "mort_hip_rng_seed");

        for (int f = 0; f < frames; f++) {
            if (f > 0) frame_input(&cam, keys, f, mouse_dx, mouse_dy);
            const double t0 = now_s();
            if (gpus > 1 && !gather_shm) {
                if ((st = mort_hip_render_gather(ctx, &cam, mode, rank == 0 ? rgba : NULL, &stats)) != MORT_OK) die(ctx, st, "mort_hip_render_gather");
            } else {
                if ((st = mort_hip_render(ctx, &cam, mode, rgba, accum, NULL, &stats)) != MORT_OK) die(ctx, st, "mort_hip_render");
                if (gpus > 1) { /* --gather shm: owned rows into the shared frame */
                    const int lr = mort_hip_local_rows(ctx, H);
                    for (int ly = 0; ly < lr; ly++) { const int y = mort_hip_global_row(ctx, ly); memcpy(shm + (size_t)y * W * 4, rgba + (size_t)y * W * 4, (size_t)W * 4); }
                }
            }
            frame_wall = now_s() - t0;
            total_ms += stats.seconds * 1e3;
            if (rank == 0) printf("Avg. time per frame: %3.1f ms\n", total_ms / (f + 1)); /* mort.cu:119 */
        }
    }

    /* ---- ranks other than 0 are done; rank 0 waits for them (their rows are in shm / were gathered) ---- */
    if (rank != 0) {
        if (ctx) mort_hip_shutdown(ctx);
        _exit(0);
    }
    int failed = 0;
    for (int r = 1; r < gpus; r++) { int ws = 0; if (waitpid(kids[r], &ws, 0) < 0 || !WIFEXITED(ws) || WEXITSTATUS(ws) != 0) failed = 1; }
    if (failed) { fprintf(stderr, "a rank failed\n"); kill_ranks(); return EXIT_FAILURE; }
    if (gpus > 1 && gather_shm) {
        const int lr = mort_hip_local_rows(ctx, H); /* rank 0's own rows are already in rgba; take the others from the mapping */
        uint8_t *own = malloc(npx * 4);
        memcpy(own, rgba, npx * 4);
        memcpy(rgba, shm, npx * 4);
        for (int ly = 0; ly < lr; ly++) { const int y = mort_hip_global_row(ctx, ly); memcpy(rgba + (size_t)y * W * 4, own + (size_t)y * W * 4, (size_t)W * 4); }
        free(own);
    }

    const double sec = (gpus > 1) ? frame_wall : stats.seconds; /* multi-GPU: wall clock of the frame on rank 0, gather included */
    printf("{\"scene\": %d, \"width\": %d, \"height\": %d, \"spp_nominal\": %d, \"spp_effective\": %d, \"depth\": %d, \"mode\": \"%s\", \"gpus\": %d, "
           "\"seconds\": %.6f, \"msamples_per_s\": %.3f, \"kernel_seconds\": %.6f, \"gather_seconds\": %.6f, \"segments\": %llu, "
           "\"algorithmic_hbm_bytes\": %llu, \"hbm_GBps\": %.4g, \"hbm_frac_of_8TBps\": %.3g, \"reference_walks\": %llu, \"kernel\": \"%s\"}\n",
           scene, W, H, cam.samples_per_pixel, eff, cam.bounce_limit, host_mode ? "host" : mode == MORT_MODE_WAVE ? "wave" : mode == MORT_MODE_THROUGHPUT ? "throughput (non-parity)" : "mega", gpus, sec,
           (double)npx * eff / sec / 1e6, stats.seconds, stats.gather_seconds, (unsigned long long)stats.segments,
           (unsigned long long)stats.algorithmic_hbm_bytes, stats.algorithmic_hbm_bytes / stats.seconds / 1e9,
           stats.algorithmic_hbm_bytes / stats.seconds / 8e12, (unsigned long long)stats.reference_walks, stats.kernel_name);
    if (out && mort_write_ppm(out, rgba, W, H) != 0) { fprintf(stderr, "cannot write %s\n", out); return EXIT_FAILURE; }
    if (dump) {
        FILE *f = fopen(dump, "wb");
        if (!f || fwrite(accum, sizeof(float), npx * 3, f) != npx * 3) { fprintf(stderr, "cannot write %s\n", dump); return EXIT_FAILURE; }
        fclose(f);
    }
    if (sout) {
        mort_rng_state *s = hstates ? hstates : malloc(npx * sizeof *s);
        if (!hstates && (st = mort_hip_rng_store(ctx, s, W, H)) != MORT_OK) die(ctx, st, "mort_hip_rng_store");
        FILE *f = fopen(sout, "wb");
        if (!f || fwrite(s, sizeof *s, npx, f) != npx) { fprintf(stderr, "cannot write %s\n", sout); return EXIT_FAILURE; }
        fclose(f);
        if (!hstates) free(s);
    }
    if (ctx) mort_hip_shutdown(ctx);
    mort_world_free(&world);
    free(texels); free(rgba); free(accum); free(hstates);
    return 0;
}
