/*
 * bvh_order_experiment2.c -- CPU model of the ordered walk that mega_bvh.h uses; tests/test_own_tree_model.py
 * runs it.
 * Own SAH tree over the reference's LEAF NODES (boxes bit-identical to the reference's), near-first walk with a stack,
 * pruned in fp32: "certain miss" (tau band) or "entered beyond closest by more than the anomaly margin"
 *      t_enter * (1 - 8e-3) - closest > E_X / |d|,  E_X = max leaf diagonal below X + 8e-3 * diagonal of X.
 * The winner (min t over every sphere hit seen, ties flagged) is checked once against the reference's box test of its
 * leaf node at t_max = t; failure or a tie sends the ray to the reference walk.  Counts steps and verifies that every
 * ray that is not sent to the reference walk agrees with it.
 *   gcc -O2 -ffp-contract=off -Iinclude -Ioracle scripts/bvh_order_experiment2.c mort_amd/csrc/host/*.c -lm -lpthread -o build/bvh_exp2
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static void seg_hook(const void *w, const void *r, float t_min);
#define MORT_ORACLE_SEGMENT_HOOK(w, r, tmin) seg_hook((w), (r), (tmin))
#include "../oracle/mort_oracle.c"
#include "mort_host.h"

typedef struct { float lo[3], hi[3]; } box3;
typedef struct { box3 b; int left, right; int leaf; float E; } onode; /* leaf >= 0: index into leaves */
static onode *T; static int nT;
static int nleaves; static box3 *lbox; static int *lnode, *lA, *lB;
static const mort_bvh *RB; static const mort_world *W;
static int opt_popcull = 0, opt_margin = 1; /* the kernel does not re-test popped nodes */

static box3 from_aabb(mort_aabb a) { box3 b = {{a.x.imin, a.y.imin, a.z.imin}, {a.x.imax, a.y.imax, a.z.imax}}; return b; }
static box3 bunion(box3 a, box3 b) { for (int k = 0; k < 3; k++) { if (b.lo[k] < a.lo[k]) a.lo[k] = b.lo[k]; if (b.hi[k] > a.hi[k]) a.hi[k] = b.hi[k]; } return a; }
static double barea(box3 a) { double x = a.hi[0] - a.lo[0], y = a.hi[1] - a.lo[1], z = a.hi[2] - a.lo[2]; return 2 * (x * y + y * z + z * x); }
static double bdiag(box3 a) { double x = a.hi[0] - a.lo[0], y = a.hi[1] - a.lo[1], z = a.hi[2] - a.lo[2]; return sqrt(x * x + y * y + z * z); }

static int cmp_axis; static int cmpf(const void *a, const void *b) {
    int i = *(const int *)a, j = *(const int *)b;
    float ci = lbox[i].lo[cmp_axis] + lbox[i].hi[cmp_axis], cj = lbox[j].lo[cmp_axis] + lbox[j].hi[cmp_axis];
    return (ci > cj) - (ci < cj);
}
static double maxleafdiag;
static int build(int *idx, int n) {
    int me = nT++;
    box3 b = lbox[idx[0]]; double ml = bdiag(lbox[idx[0]]);
    for (int i = 1; i < n; i++) { b = bunion(b, lbox[idx[i]]); double d = bdiag(lbox[idx[i]]); if (d > ml) ml = d; }
    T[me].b = b; T[me].leaf = -1; T[me].E = nextafterf((float)(ml * 1.001 + 8e-3 * bdiag(b)), INFINITY);
    if (n == 1) { T[me].leaf = idx[0]; return me; }
    double best = 1e300; int bax = 0, bsplit = n / 2;
    int *tmp = malloc(n * sizeof(int)); double *ra = malloc(n * sizeof(double));
    for (int ax = 0; ax < 3; ax++) {
        memcpy(tmp, idx, n * sizeof(int)); cmp_axis = ax; qsort(tmp, n, sizeof(int), cmpf);
        box3 r = lbox[tmp[n - 1]]; ra[n - 1] = barea(r);
        for (int i = n - 2; i >= 0; i--) { r = bunion(r, lbox[tmp[i]]); ra[i] = barea(r); }
        box3 l = lbox[tmp[0]];
        for (int i = 1; i < n; i++) { double c = barea(l) * i + ra[i] * (n - i); if (c < best) { best = c; bax = ax; bsplit = i; } l = bunion(l, lbox[tmp[i]]); }
    }
    cmp_axis = bax; qsort(idx, n, sizeof(int), cmpf);
    free(tmp); free(ra);
    int l = build(idx, bsplit), r = build(idx + bsplit, n - bsplit);
    T[me].left = l; T[me].right = r;
    return me;
}

static unsigned long long n_rays, r_box, r_sph, r_leaf, v_steps, v_leaf, v_sph, v_fallback, v_mismatch, v_popcull, v_maxsp, hist_steps[64];

/* mega_bvh.h own_prune, operation for operation (p = fma(b, inv, -o*inv); tau = 2^-20 max(|te|,|tx|) + band) */
static float g_band, g_m[3];
static int prune(const onode *nd, const float o[3], const float inv[3], float invlen, float closest, float *key) {
    float te = 0.001f, tx = INFINITY;
    (void)o;
    for (int k = 0; k < 3; k++) {
        float p0 = fmaf(nd->b.lo[k], inv[k], -g_m[k]), p1 = fmaf(nd->b.hi[k], inv[k], -g_m[k]);
        te = fmaxf(te, fminf(p0, p1)); tx = fminf(tx, fmaxf(p0, p1));
    }
    const float tau = fmaf(fmaxf(fabsf(te), fabsf(tx)), 9.5367431640625e-07f, g_band);
    const float k2 = opt_margin ? fmaf(te, 0.992f, -fmaf(nd->E, invlen, tau)) : te - tau;
    *key = te;
    return (tx - te < -tau) || (k2 > closest);
}
static float sph_t(const mort_sphere *s, const ray *r, float t_min, float t_max, int *ok) {
    hit_record rec; memset(&rec, 0, sizeof rec);
    *ok = sphere_hit(s, r, t_min, t_max, &rec);
    return rec.t;
}
static void ref_walk(const ray *r, float t_min, int node, float *closest, int *best) {
    r_box++;
    if (!aabb_hit(&RB->bounding_boxes[node], r, t_min, *closest)) return;
    if (!RB->is_internal_node[node]) {
        int ok; float t; r_leaf++;
        r_sph++; t = sph_t(&W->objs.host_sphere[RB->left_children_idxs[node]], r, t_min, *closest, &ok); if (ok) { *closest = t; *best = RB->left_children_idxs[node]; }
        if (RB->right_children_idxs[node] != RB->left_children_idxs[node]) r_sph++;
        t = sph_t(&W->objs.host_sphere[RB->right_children_idxs[node]], r, t_min, *closest, &ok); if (ok) { *closest = t; *best = RB->right_children_idxs[node]; }
        return;
    }
    ref_walk(r, t_min, RB->left_children_idxs[node], closest, best);
    ref_walk(r, t_min, RB->right_children_idxs[node], closest, best);
}
static void seg_hook(const void *wv, const void *rv, float t_min) {
    const ray *r = rv; (void)wv;
    n_rays++;
    float c_ref = INFINITY; int b_ref = -1;
    ref_walk(r, t_min, 0, &c_ref, &b_ref);
    const float o[3] = {r->orig.e[0], r->orig.e[1], r->orig.e[2]};
    float inv[3]; int deg = 0;
    for (int k = 0; k < 3; k++) { inv[k] = 1.0f / r->dir.e[k]; float a = fabsf(inv[k]); if (!(a > 1e-15f && a < 1e15f)) deg = 1; g_m[k] = o[k] * inv[k]; }
    const float mm = fmaxf(fmaxf(fabsf(g_m[0]), fabsf(g_m[1])), fabsf(g_m[2]));
    if (deg || !(mm < 1e30f)) { v_fallback++; return; }
    g_band = mm * 4.76837158203125e-07f;
    const float invlen = 1.01f / sqrtf(vlen2(r->dir));
    float cl = INFINITY; int bs = -1, bl = -1, tie = 0; int stk[64]; float stt[64]; int n = 0, c = 0; float key;
    unsigned long long steps = 0;
    if (prune(&T[0], o, inv, invlen, cl, &key)) c = -1;
    while (c >= 0) {
        if (T[c].leaf >= 0) {
            const int l = T[c].leaf; v_leaf++;
            for (int k = 0; k < 2; k++) {
                const int s = k ? lB[l] : lA[l]; if (k && lA[l] == lB[l]) break;
                int ok; v_sph++;
                float t = sph_t(&W->objs.host_sphere[s], r, t_min, cl, &ok);
                if (ok) { if (t == cl && bs >= 0) tie = 1; cl = t; bs = s; bl = l; }
            }
            c = -1;
            while (n) { n--; if (opt_popcull && stt[n] > cl) { v_popcull++; continue; } c = stk[n]; break; }
            continue;
        }
        steps++;
        float kl, kr; const int L = T[c].left, R = T[c].right;
        const int pl = prune(&T[L], o, inv, invlen, cl, &kl), pr = prune(&T[R], o, inv, invlen, cl, &kr);
        if (pl && pr) { c = -1; while (n) { n--; if (opt_popcull && stt[n] > cl) { v_popcull++; continue; } c = stk[n]; break; } }
        else if (pl) c = R;
        else if (pr) c = L;
        else { if (kl <= kr) { stk[n] = R; stt[n++] = kr; c = L; } else { stk[n] = L; stt[n++] = kl; c = R; } if ((unsigned long long)n > v_maxsp) v_maxsp = n; }
    }
    v_steps += steps; hist_steps[steps < 63 ? steps : 63]++;
    if (tie || (bs >= 0 && !aabb_hit(&RB->bounding_boxes[lnode[bl]], r, t_min, cl))) v_fallback++;
    else if (bs != b_ref || (bs >= 0 && cl != c_ref)) v_mismatch++;
}
int main(int argc, char **argv) {
    int width = argc > 1 ? atoi(argv[1]) : 300, spp = argc > 2 ? atoi(argv[2]) : 4;
    const int scene = argc > 5 ? atoi(argv[5]) : 1;
    if (argc > 3) opt_margin = atoi(argv[3]);
    if (argc > 4) opt_popcull = atoi(argv[4]);
    static mort_world w; mort_camera cam;
    mort_world_init(&w);
    mort_scene_opts opts; memset(&opts, 0, sizeof opts);
    mort_scene_build(scene, &w, &cam, &opts);
    cam.image_width = width; cam.samples_per_pixel = spp;
    mort_camera_initialize(&cam);
    W = &w; RB = &w.objs.host_bvh[0];
    nleaves = 0; lbox = malloc(4096 * sizeof(box3)); lnode = malloc(4096 * sizeof(int)); lA = malloc(4096 * sizeof(int)); lB = malloc(4096 * sizeof(int));
    int st[128], sp = 0; st[sp++] = 0;
    while (sp) {
        int n = st[--sp];
        if (RB->is_internal_node[n]) { st[sp++] = RB->right_children_idxs[n]; st[sp++] = RB->left_children_idxs[n]; continue; }
        lbox[nleaves] = from_aabb(RB->bounding_boxes[n]); lnode[nleaves] = n; lA[nleaves] = RB->left_children_idxs[n]; lB[nleaves] = RB->right_children_idxs[n]; nleaves++;
    }
    T = malloc(2 * nleaves * sizeof(onode)); nT = 0;
    int *idx = malloc(nleaves * sizeof(int)); for (int i = 0; i < nleaves; i++) idx[i] = i;
    build(idx, nleaves);
    const int H = cam.image_height;
    mort_rng_state *states = malloc((size_t)width * H * sizeof *states);
    mort_oracle_rng_seed(states, 1984, width, H);
    uint8_t *rgba = malloc((size_t)width * H * 4);
    mort_oracle_stats stats;
    mort_oracle_render(&w, &cam, states, 0, H, rgba, NULL, NULL, 1, &stats);
    printf("ref leaves %d own nodes %d rays %llu margin %d popcull %d\n", nleaves, nT, n_rays, opt_margin, opt_popcull);
    printf("reference walk: box tests/ray %.2f leaves/ray %.2f sphere tests/ray %.2f\n", (double)r_box / n_rays, (double)r_leaf / n_rays, (double)r_sph / n_rays);
    printf("ordered walk: node steps/ray %.2f leaves/ray %.2f sphere tests/ray %.2f popculls/ray %.2f max stack %llu fallbacks %llu mismatches %llu\n",
           (double)v_steps / n_rays, (double)v_leaf / n_rays, (double)v_sph / n_rays, (double)v_popcull / n_rays, v_maxsp, v_fallback, v_mismatch);
    printf("steps histogram:"); for (int i = 0; i < 64; i++) printf(" %llu", hist_steps[i]); printf("\n");
    return v_mismatch != 0;
}
