/*
 * host_render.hip -- `mort --mode host`: the render path as a straight host loop of the SAME kernel body the GPU runs
 * (dev_pixel.h render_pixel: Camera::render / ray_color / get_ray, camera.cuh:86-242, over world::hit, materials, pdfs,
 * textures and XORWOW exactly as compiled for the device, -ffp-contract=off on both sides), on host threads.
 *
 * This is BASELINE config 1 ("host-side serial C render loop", Scene 1 200x112, 4 spp) and the CPU figure the
 * north_star wants timed beside the GPU numbers (T = 1 and T = all cores).  It is an explicit mode of the CLI / ABI,
 * never a fallback: mort_hip_render / mort_hip_render_device fail when there is no MI355X.  It needs no GPU and no
 * HIP runtime call.  It is product code and does not touch oracle/ (the test-side checker).
 *
 * Replaces, on the host: world::toDevice() + setup_rng + renderKernel (world.cuh:98-102, rng.cuh:8-15, mort.cu:44-47).
 */
#include <hip/hip_runtime.h>

#include <pthread.h>
#include <time.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mort_hip.h"
#include "dev_pixel.h"
#include "scene_blob.h"
#include "seed_host.h"

#pragma clang fp contract(off)

namespace {

struct HostJob {
    RenderArgs a;
    GenWalk gw;
    bool tree;
    std::atomic<int> next_row{0};
    int row_end = 0;
    std::atomic<unsigned long long> segments{0}, draws{0}, scans{0};
};

void *host_worker(void *p) {
    HostJob *j = (HostJob *)p;
    unsigned long long seg = 0, drw = 0, scn = 0;
    for (;;) {
        const int ly = j->next_row.fetch_add(1);
        if (ly >= j->row_end) break;
        for (int x = 0; x < j->a.width; x++) {
            const PixelTotals t = j->tree ? render_pixel<true>(j->a, &j->gw, x, ly, &scn) : render_pixel<false>(j->a, nullptr, x, ly, nullptr);
            seg += t.segments; drw += t.draws;
        }
    }
    j->segments += seg; j->draws += drw; j->scans += scn;
    if (std::getenv("MORT_HOST_DEBUG")) std::fprintf(stderr, "[host worker %lu] %llu segments\n", (unsigned long)pthread_self(), seg);
    return nullptr;
}

double now_s() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

} // namespace

extern "C" int mort_hip_rng_seed_host(uint64_t seed, int width, int height, mort_rng_state *states) {
    if (!states || width <= 0 || height <= 0) return MORT_ERR_INVALID;
    static std::vector<XMat> seq; /* 32 matrices of 3.2 KB, built once */
    static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    pthread_mutex_lock(&mu);
    if (seq.empty()) build_seq_matrices(seq);
    pthread_mutex_unlock(&mu);
    const SeedWords sw = seed_scramble(seed);
    const size_t n = (size_t)width * (size_t)height;
    for (size_t i = 0; i < n; i++) {
        mort_rng_state st;
        std::memset(&st, 0, sizeof st);
        st.d = sw.d;
        seed_subsequence(seq, sw, (unsigned long long)i, st.v);
        states[i] = st;
    }
    return MORT_OK;
}

extern "C" int mort_hip_render_host(const mort_world *world, const mort_camera *cam, mort_rng_state *states, int nthreads, int flags,
                                    uint8_t *rgba_out, float *accum_out, uint32_t *segments_px_out, mort_stats *stats) {
    if (!world || !cam || !states || !rgba_out) return MORT_ERR_INVALID;
    const int W = cam->image_width, H = cam->image_height;
    if (W <= 0 || H <= 0 || cam->sqrt_spp < 0) return MORT_ERR_INVALID;
    if (cam->bounce_limit < 0 || cam->bounce_limit > MORT_MAX_BOUNCE_LIMIT) return MORT_ERR_CAPACITY;
    SceneBlob sb;
    int st = build_scene_blob(world, sb);
    if (st != MORT_OK) return st;
    if ((st = check_light_object(sb.comp, world->objs.num_hittable_list, cam->light_obj_type, cam->light_obj_idx)) != MORT_OK) return st;

    HostJob job;
    std::memset(&job.a, 0, sizeof job.a);
    scene_view(sb, sb.bytes.data(), job.a.sc); /* the same flat arrays the kernels read, in host memory */
    render_args_camera(job.a, cam);
    job.a.rank = 0; job.a.nranks = 1; job.a.rows_per_block = 8; job.a.local_rows = H;
    job.a.states = states;
    job.a.rgba = (uchar4 *)rgba_out; job.a.accum = accum_out; job.a.seg_px = segments_px_out;
    job.a.counters = nullptr;
    /* the unified tree walked by one lane (dev_gen.h gen_world_hit), when the world has one and the caller asks for it */
    const mortc::Compiled &o = sb.comp;
    job.tree = (flags & MORT_HOST_TREE) && o.g_ok;
    if (job.tree) {
        for (int k = 0; k < 3; k++) {
            const float v = cam->center.e[k];
            if (!(v >= o.g_lo[k] - o.g_reach && v <= o.g_hi[k] + o.g_reach)) job.tree = false;
        }
    }
    std::memset(&job.gw, 0, sizeof job.gw);
    if (job.tree) {
        job.gw.nodes = o.g_nodes.data(); job.gw.entries = o.g_entries.data(); job.gw.chains = o.g_chains.data();
        job.gw.ranks = o.g_ranks.data(); job.gw.n_spheres = (int)o.spheres.size();
        job.gw.n_chains = (int)(o.g_chains.size() / 2); job.gw.root = o.g_root; job.gw.first_medium = o.g_first_medium;
        job.gw.gx = o.g_c[0]; job.gw.gy = o.g_c[1]; job.gw.gz = o.g_c[2]; job.gw.gR = o.g_R; job.gw.mnear = o.g_mnear; job.gw.kmin = o.g_kmin;
    }

    job.row_end = H;
    if (const char *rows = std::getenv("MORT_HOST_ROWS")) { /* debug aid: "y0,y1" renders rows [y0, y1) only */
        int y0 = 0, y1 = H;
        if (std::sscanf(rows, "%d,%d", &y0, &y1) == 2 && y0 >= 0 && y1 <= H && y0 < y1) { job.next_row = y0; job.row_end = y1; }
    }
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    const double t0 = now_s();
    std::vector<pthread_t> th((size_t)nthreads - 1);
    size_t started = 0;
    for (; started < th.size(); started++)
        if (pthread_create(&th[started], nullptr, host_worker, &job) != 0) break;
    host_worker(&job);
    for (size_t i = 0; i < started; i++) pthread_join(th[i], nullptr);
    const double dt = now_s() - t0;

    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        stats->seconds = dt;
        stats->segments = job.segments; stats->rng_draws = job.draws;
        stats->pixels = (uint64_t)W * (uint64_t)H;
        stats->eff_samples = stats->pixels * (uint64_t)(cam->sqrt_spp * cam->sqrt_spp);
        stats->algorithmic_hbm_bytes = 0;
        stats->local_rows = H;
        stats->reference_walks = job.scans;
        std::snprintf(stats->kernel_name, sizeof stats->kernel_name, "host loop, %d thread(s), %s", (int)started + 1, job.tree ? "unified tree" : "item scan");
    }
    return MORT_OK;
}
