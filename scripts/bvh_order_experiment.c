/*
 * bvh_order_experiment.c -- CPU study for the ordered-traversal design: for every segment the oracle
 * traces in a small Scene 1 render, compare
 *   (R) the reference's left-first walk of its own BVH (aabb::hit per node, both leaf spheres), with
 *   (O) a near-first walk of this build's own SAH tree over the same spheres, pruned conservatively in fp32, where a
 *       candidate hit counts only if the reference's box test of its reference LEAF NODE passes at t_max = t
 *       ("member"), and a hit whose leaf box passes at t_max = inf but not at t ("anomaly") sends the ray to (R).
 * Prints step counts for both and the number of rays where (O) without fallback differs from (R) (must be 0).
 *
 *   gcc -O2 -ffp-contract=off -Iinclude -Ioracle scripts/bvh_order_experiment.c mort_amd/csrc/host/*.c -lm -lpthread -o build/bvh_exp
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
struct ray_s;
static void seg_hook(const void *w, const void *r, float t_min);
#define MORT_ORACLE_SEGMENT_HOOK(w, r, tmin) seg_hook((w), (r), (tmin))
#include "../oracle/mort_oracle.c"
#include "mort_host.h"

typedef struct { float lo[3], hi[3]; } box3;
typedef struct { box3 b; int left, right; int prim; } onode; /* prim >= 0: leaf */
static onode *T; static int nT;
static int nprims; static box3 *pbox; static int *pleaf /* ref leaf node */, *prank, *psphere;
static const mort_bvh *RB; static const mort_world *W;

static box3 from_aabb(mort_aabb a) { box3 b = {{a.x.imin, a.y.imin, a.z.imin}, {a.x.imax, a.y.imax, a.z.imax}}; return b; }
static box3 bunion(box3 a, box3 b) { for (int k = 0; k < 3; k++) { if (b.lo[k] < a.lo[k]) a.lo[k] = b.lo[k]; if (b.hi[k] > a.hi[k]) a.hi[k] = b.hi[k]; } return a; }
static double barea(box3 a) { double x = a.hi[0] - a.lo[0], y = a.hi[1] - a.lo[1], z = a.hi[2] - a.lo[2]; return 2 * (x * y + y * z + z * x); }

static int cmp_axis; static int cmpf(const void *a, const void *b) {
    int i = *(const int *)a, j = *(const int *)b;
    float ci = pbox[i].lo[cmp_axis] + pbox[i].hi[cmp_axis], cj = pbox[j].lo[cmp_axis] + pbox[j].hi[cmp_axis];
    return (ci > cj) - (ci < cj);
}
static int build(int *idx, int n) {
    int me = nT++;
    box3 b = pbox[idx[0]]; for (int i = 1; i < n; i++) b = bunion(b, pbox[idx[i]]);
    T[me].b = b; T[me].prim = -1;
    if (n == 1) { T[me].prim = idx[0]; return me; }
    double best = 1e300; int bax = 0, bsplit = n / 2;
    int *tmp = malloc(n * sizeof(int)); double *ra = malloc(n * sizeof(double));
    for (int ax = 0; ax < 3; ax++) {
        memcpy(tmp, idx, n * sizeof(int)); cmp_axis = ax; qsort(tmp, n, sizeof(int), cmpf);
        box3 r = pbox[tmp[n - 1]]; ra[n - 1] = barea(r);
        for (int i = n - 2; i >= 0; i--) { r = bunion(r, pbox[tmp[i]]); ra[i] = barea(r); }
        box3 l = pbox[tmp[0]];
        for (int i = 1; i < n; i++) { double c = barea(l) * i + ra[i] * (n - i); if (c < best) { best = c; bax = ax; bsplit = i; } l = bunion(l, pbox[tmp[i]]); }
    }
    cmp_axis = bax; qsort(idx, n, sizeof(int), cmpf);
    free(tmp); free(ra);
    int l = build(idx, bsplit), r = build(idx + bsplit, n - bsplit);
    T[me].left = l; T[me].right = r;
    return me;
}

/* counters */
static unsigned long long v_steps, v_leaf, v_fallback, v_mismatch, v_popcull, hist_steps[64];
static unsigned long long n_rays, r_box, r_sph, o_steps, o_box, o_leaf, o_sph, o_exact, o_anom, o_fallback, o_mismatch, o_maxstack;

/* conservative fp32 prune (dev_trace.h accel_prune) returning entry distance */
static int prune(const box3 *b, const float o[3], const float inv[3], float closest, float *t_enter) {
    float te = 0.001f, tx = closest * 1.002f + 1e-3f, m = 0;
    for (int k = 0; k < 3; k++) {
        float p0 = (b->lo[k] - o[k]) * inv[k], p1 = (b->hi[k] - o[k]) * inv[k];
        te = fmaxf(te, fminf(p0, p1)); tx = fminf(tx, fmaxf(p0, p1));
        m = fmaxf(m, fmaxf(fabsf(p0), fabsf(p1)));
    }
    *t_enter = te;
    return (m < 1e30f) && (tx - te < -(m * 9.5367431640625e-07f));
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
        int ok; float t;
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
    for (int k = 0; k < 3; k++) { inv[k] = (float)(1.0 / (double)r->dir.e[k]); float a = fabsf(inv[k]); if (!(a > 1e-30f && a < 1e30f)) deg = 1; }
    if (deg) { o_fallback++; return; }
    float closest = INFINITY, anom_t = INFINITY; int best = -1, best_rank = -1;
    int stack[64], sp = 0, cur = 0; float te;
    if (prune(&T[0].b, o, inv, closest, &te)) cur = -1;
    unsigned long long maxsp = 0;
    while (cur >= 0) {
        if (T[cur].prim >= 0) {
            const int p = T[cur].prim; int ok;
            o_leaf++; o_sph++;
            float t = sph_t(&W->objs.host_sphere[psphere[p]], r, t_min, closest, &ok);
            if (ok && (t < closest || prank[p] > best_rank)) {
                o_exact++;
                if (aabb_hit(&RB->bounding_boxes[pleaf[p]], r, t_min, t)) { closest = t; best = psphere[p]; best_rank = prank[p]; }
                else if (aabb_hit(&RB->bounding_boxes[pleaf[p]], r, t_min, INFINITY)) { if (t < anom_t) anom_t = t;
                    static int shown; if (shown < 12) { shown++; const mort_aabb *bb = &RB->bounding_boxes[pleaf[p]];
                        printf("anom sphere %d r=%g t=%.9g o=(%g %g %g) d=(%g %g %g) box y [%g %g] x [%g %g]\n", psphere[p], W->objs.host_sphere[psphere[p]].radius, t, o[0], o[1], o[2], r->dir.e[0], r->dir.e[1], r->dir.e[2], bb->y.imin, bb->y.imax, bb->x.imin, bb->x.imax); } }
            }
            cur = sp ? stack[--sp] : -1;
            continue;
        }
        o_steps++; o_box += 2;
        float tl, tr; const int L = T[cur].left, R = T[cur].right;
        const int pl = prune(&T[L].b, o, inv, closest, &tl), pr = prune(&T[R].b, o, inv, closest, &tr);
        if (pl && pr) cur = sp ? stack[--sp] : -1;
        else if (pl) cur = R;
        else if (pr) cur = L;
        else { if (tl <= tr) { stack[sp++] = R; cur = L; } else { stack[sp++] = L; cur = R; } if ((unsigned long long)sp > maxsp) maxsp = sp; }
    }
    if (maxsp > o_maxstack) o_maxstack = maxsp;
    { /* variant V: min over all visited sphere hits (ties flag), one membership test at the end; stack keeps entry distances */
        float cl = INFINITY; int bs = -1, bp = -1, tie = 0; int stk[64]; float stt[64]; int n = 0, c = 0; float te2;
        unsigned long long steps = 0;
        if (prune(&T[0].b, o, inv, cl, &te2)) c = -1;
        while (c >= 0) {
            if (T[c].prim >= 0) {
                const int p = T[c].prim; int ok; v_leaf++;
                float t = sph_t(&W->objs.host_sphere[psphere[p]], r, t_min, cl, &ok);
                if (ok) { if (t == cl && bs >= 0) tie = 1; cl = t; bs = psphere[p]; bp = p; }
                c = -1;
                while (n) { n--; if (stt[n] > cl * 1.002f + 1e-3f) { v_popcull++; continue; } c = stk[n]; break; }
                continue;
            }
            steps++;
            float tl, tr; const int L = T[c].left, R = T[c].right;
            const int pl = prune(&T[L].b, o, inv, cl, &tl), pr = prune(&T[R].b, o, inv, cl, &tr);
            if (pl && pr) { c = -1; while (n) { n--; if (stt[n] > cl * 1.002f + 1e-3f) { v_popcull++; continue; } c = stk[n]; break; } }
            else if (pl) c = R;
            else if (pr) c = L;
            else { if (tl <= tr) { stk[n] = R; stt[n++] = tr; c = L; } else { stk[n] = L; stt[n++] = tl; c = R; } }
        }
        v_steps += steps; hist_steps[steps < 63 ? steps : 63]++;
        if (tie || (bs >= 0 && !aabb_hit(&RB->bounding_boxes[pleaf[bp]], r, t_min, cl))) v_fallback++;
        else if (bs != b_ref || (bs >= 0 && cl != c_ref)) v_mismatch++;
    }
    if (anom_t < INFINITY && anom_t <= closest) { o_anom++; o_fallback++; return; }
    if (best != b_ref || (best >= 0 && closest != c_ref)) o_mismatch++;
}

int main(int argc, char **argv) {
    int width = argc > 1 ? atoi(argv[1]) : 300, spp = argc > 2 ? atoi(argv[2]) : 4;
    static mort_world w; mort_camera cam;
    mort_world_init(&w);
    mort_scene_opts opts; memset(&opts, 0, sizeof opts);
    mort_scene_build(1, &w, &cam, &opts);
    cam.image_width = width; cam.samples_per_pixel = spp;
    mort_camera_initialize(&cam);
    W = &w; RB = &w.objs.host_bvh[0];
    /* primitives in reference order: pre-order over leaves, left then right */
    nprims = 0; pbox = malloc(4096 * sizeof(box3)); pleaf = malloc(4096 * sizeof(int)); prank = malloc(4096 * sizeof(int)); psphere = malloc(4096 * sizeof(int));
    int st[128], sp = 0; st[sp++] = 0;
    while (sp) {
        int n = st[--sp];
        if (RB->is_internal_node[n]) { st[sp++] = RB->right_children_idxs[n]; st[sp++] = RB->left_children_idxs[n]; continue; }
        int a = RB->left_children_idxs[n], b = RB->right_children_idxs[n];
        for (int k = 0; k < 2; k++) {
            int s = k ? b : a; if (k && a == b) break;
            pbox[nprims] = from_aabb(mort_get_bbox(&w, MORT_OBJ_SPHERE, s)); pleaf[nprims] = n; prank[nprims] = nprims; psphere[nprims] = s; nprims++;
        }
    }
    T = malloc(2 * nprims * sizeof(onode)); nT = 0;
    int *idx = malloc(nprims * sizeof(int)); for (int i = 0; i < nprims; i++) idx[i] = i;
    build(idx, nprims);
    const int H = cam.image_height;
    mort_rng_state *states = malloc((size_t)width * H * sizeof *states);
    mort_oracle_rng_seed(states, 1984, width, H);
    uint8_t *rgba = malloc((size_t)width * H * 4);
    mort_oracle_stats stats;
    mort_oracle_render(&w, &cam, states, 0, H, rgba, NULL, NULL, 1, &stats);
    printf("prims %d own nodes %d rays %llu\n", nprims, nT, n_rays);
    printf("reference walk: box tests/ray %.2f sphere tests/ray %.2f\n", (double)r_box / n_rays, (double)r_sph / n_rays);
    printf("ordered walk:   node steps/ray %.2f (box tests %.2f) leaves/ray %.2f exact leaf-box tests/ray %.3f max stack %llu\n",
           (double)o_steps / n_rays, (double)o_box / n_rays, (double)o_leaf / n_rays, (double)o_exact / n_rays, o_maxstack);
    printf("anomalies %llu fallbacks %llu (%.4f%%) mismatches %llu\n", o_anom, o_fallback, 100.0 * o_fallback / n_rays, o_mismatch);
    printf("variant V (verify winner only, pop-time cull): node steps/ray %.2f leaves/ray %.2f popculls/ray %.2f fallbacks %llu mismatches %llu\n",
           (double)v_steps / n_rays, (double)v_leaf / n_rays, (double)v_popcull / n_rays, v_fallback, v_mismatch);
    printf("steps histogram:"); for (int i = 0; i < 64; i++) printf(" %llu", hist_steps[i]); printf("\n");
    return o_mismatch != 0 || v_mismatch != 0;
}
