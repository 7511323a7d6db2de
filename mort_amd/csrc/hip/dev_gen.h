/*
 * dev_gen.h -- the pieces of the unified-tree closest-hit search (scene_compile.h build_unified) that every form of it
 * shares: the state-machine megakernel (mega_gen.hip), the wavefront traversal kernel (wave_gen.hip) and the single-lane
 * walk that the host loop and the CPU tests run (gen_world_hit below).  One body per step, so the forms cannot drift.
 *
 * Replaces the brute-force part of world::hit (world.cuh:122-168) for worlds without reference BVHs; the proof that the
 * tree walk returns the scan's primitive is in the header of build_unified.
 */
#ifndef MORT_DEV_GEN_H
#define MORT_DEV_GEN_H

#include "dev_render.h"

#pragma clang fp contract(off)

#define GBEST_NONE 0xffffffffu
#define GCHAIN_MEDIUM 0x7fu /* best = GENT(0, GCHAIN_MEDIUM, item index): the hit is a constant medium */
enum { GFL_REF = 2 }; /* the tree walk does not decide this ray: the scan does */

/* the twelve planes of a node's two child boxes (dev_scene.h DNodeQ): plane = fmaf(q, step, origin), exactly as the builder checked it */
struct GenBoxes { float x0min, x0max, y0min, y0max, z0min, z0max, x1min, x1max, y1min, y1max, z1min, z1max; uint32_t c0, c1; };
DEV float gen_bits_float(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f; __builtin_memcpy(&f, &u, 4); return f;
#endif
}
DEV GenBoxes gen_node_decode(float ox, float oy, float oz, uint32_t exps, uint32_t q0, uint32_t q1, uint32_t q2, uint32_t children) {
    const float sx = gen_bits_float((exps & 0xffu) << 23), sy = gen_bits_float(((exps >> 8) & 0xffu) << 23), sz = gen_bits_float(((exps >> 16) & 0xffu) << 23);
    GenBoxes b;
    b.x0min = __builtin_fmaf((float)(q0 & 0xffu), sx, ox); b.x0max = __builtin_fmaf((float)((q0 >> 8) & 0xffu), sx, ox);
    b.y0min = __builtin_fmaf((float)((q0 >> 16) & 0xffu), sy, oy); b.y0max = __builtin_fmaf((float)(q0 >> 24), sy, oy);
    b.z0min = __builtin_fmaf((float)(q1 & 0xffu), sz, oz); b.z0max = __builtin_fmaf((float)((q1 >> 8) & 0xffu), sz, oz);
    b.x1min = __builtin_fmaf((float)((q1 >> 16) & 0xffu), sx, ox); b.x1max = __builtin_fmaf((float)(q1 >> 24), sx, ox);
    b.y1min = __builtin_fmaf((float)(q2 & 0xffu), sy, oy); b.y1max = __builtin_fmaf((float)((q2 >> 8) & 0xffu), sy, oy);
    b.z1min = __builtin_fmaf((float)((q2 >> 16) & 0xffu), sz, oz); b.z1max = __builtin_fmaf((float)(q2 >> 24), sz, oz);
    b.c0 = children & 0xffffu; b.c1 = children >> 16;
    return b;
}

/* what a walk needs besides the scene tables */
struct GenWalk {
    const DNodeQ *nodes; const uint32_t *entries; const int *chains;
    const uint32_t *ranks; /* scan-order rank of every primitive: [sphere index] then [n_spheres + quad index] (ties only) */
    int n_spheres;
    int n_chains; uint32_t root; int first_medium;
    float gx, gy, gz, gR, mnear, kmin;
};

/* per-ray constants of the box test: p = b * inv - o * inv */
struct GenRay { float ix, iy, iz, mx, my, mz, band; };

DEV bool gen_inv_ok(float v) { const float a = mort_fabsf(v); return a > 1e-15f && a < 1e15f; } /* !(NaN) too */

/* Returns false for rays the tree walk must not decide (a reciprocal direction component that is zero, denormal, huge or
 * NaN): the scan decides those.  band = 2^-21 max |o * inv|  (+ the far-origin widening: a sphere test's false-positive
 * zone grows with |oc|^2, build_unified). */
DEV bool gen_ray_setup(const Ray &ray, float gx, float gy, float gz, float gR, float mnear, float kmin, GenRay &gr) {
    gr.ix = 1.0f / ray.d.x; gr.iy = 1.0f / ray.d.y; gr.iz = 1.0f / ray.d.z;
    gr.mx = ray.o.x * gr.ix; gr.my = ray.o.y * gr.iy; gr.mz = ray.o.z * gr.iz;
    const float mm = __builtin_fmaxf(__builtin_fmaxf(mort_fabsf(gr.mx), mort_fabsf(gr.my)), mort_fabsf(gr.mz));
    gr.band = mm * 4.76837158203125e-07f; /* 2^-21 */
    const float ex = ray.o.x - gx, ey = ray.o.y - gy, ez = ray.o.z - gz;
    const float far = mort_sqrtf(ex * ex + ey * ey + ez * ez) * 1.000001f + gR;
    const float mi = __builtin_fmaxf(__builtin_fmaxf(mort_fabsf(gr.ix), mort_fabsf(gr.iy)), mort_fabsf(gr.iz));
    if (far > mnear) gr.band = __builtin_fmaf(kmin * far * far, mi, gr.band);
    return gen_inv_ok(gr.ix) && gen_inv_ok(gr.iy) && gen_inv_ok(gr.iz) && (mm < 1e30f);
}

/* true = skip the box.  te: entry parameter (>= t_min = 0.001).  The fp32 slab values differ from exact arithmetic on
 * the (padded) box by <= 2^-21 |p| + 2^-24 |o * inv|; tau = 2^-20 max(|te|, |tx|) + band covers both ends.  Skip only if
 * the ray leaves the box before it enters it, or enters it later than closest_so_far, by more than tau. */
DEV bool gen_prune(float xmin, float xmax, float ymin, float ymax, float zmin, float zmax, const GenRay &r, float closest, float &te_out) {
    const float px0 = __builtin_fmaf(xmin, r.ix, -r.mx), px1 = __builtin_fmaf(xmax, r.ix, -r.mx);
    const float py0 = __builtin_fmaf(ymin, r.iy, -r.my), py1 = __builtin_fmaf(ymax, r.iy, -r.my);
    const float pz0 = __builtin_fmaf(zmin, r.iz, -r.mz), pz1 = __builtin_fmaf(zmax, r.iz, -r.mz);
    const float te = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(px0, px1), __builtin_fminf(py0, py1)),
                                     __builtin_fmaxf(__builtin_fminf(pz0, pz1), 0.001f));
    const float tx = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(px0, px1), __builtin_fmaxf(py0, py1)), __builtin_fmaxf(pz0, pz1));
    const float tau = __builtin_fmaf(__builtin_fmaxf(mort_fabsf(te), mort_fabsf(tx)), 9.5367431640625e-07f, r.band);
    te_out = te;
    return (tx - te < -tau) || (te - tau > closest);
}

/* sphere::hit (objects.cuh:60-77) returning the accepted root or -1 (an accepted root is >= t_min > 0 or NaN) */
DEV float gen_sphere_root(const DSphere &s, const Ray &r, float a, float t_min, float t_max) {
    const V3 oc = vsub(r.o, sphere_center(s, r.tm));
    const float half_b = vdot(oc, r.d);
    const float c = vlen2(oc) - s.radius * s.radius;
    const float discriminant = half_b * half_b - a * c;
    if (discriminant < 0) return -1.0f;
    const float sqrtd = mort_sqrtf(discriminant);
    float root = (-half_b - sqrtd) / a;
    if (root < t_min || t_max < root) {
        root = (-half_b + sqrtd) / a;
        if (root < t_min || t_max < root) return -1.0f;
    }
    return root;
}

/* scan-order rank of an entry (world.cuh:122-168 visits items in order, primitives of an item in index order) */
DEV uint32_t gen_rank(const uint32_t *ranks, int n_spheres, uint32_t e) { return ranks[(GENT_QUAD(e) ? (uint32_t)n_spheres : 0u) + GENT_IDX(e)]; }

/* one primitive of a leaf: its own hit test in its own frame (sphere::hit, quad::hit :190-215, under translate::hit /
 * rotate_y::hit :268-278,334-366), t_max = closest_so_far.  Equal t: the scan keeps whichever it meets LAST (both tests
 * accept t == t_max), so the primitive with the higher scan rank stays -- every primitive whose own t equals the final
 * minimum is visited (a box is pruned only if it is entered strictly later than closest_so_far), so this is the scan's
 * choice exactly.  A NaN root is flagged: the scan itself decides that ray. */
DEV void gen_leaf_test(const DScene &sc, const int *chains, const uint32_t *ranks, int n_spheres, const DSphere *spheres, const DQuad *quads, uint32_t e,
                       const Ray &ray, float ray_a, float &closest, uint32_t &best, int &flags) {
    const uint32_t cid = GENT_CHAIN(e), idx = GENT_IDX(e);
    Ray r = ray;
    float ra = ray_a;
    if (cid != 0) {
        r = apply_chain(sc, ray, chains[2 * cid], chains[2 * cid + 1]);
        ra = vlen2(r.d);
    }
    float t;
    if (GENT_QUAD(e)) {
        float al, be;
        if (!quad_hit_t(quads[idx], r, 0.001f, closest, t, al, be)) t = -1.0f;
    } else {
        t = gen_sphere_root(spheres[idx], r, ra, 0.001f, closest);
    }
    if (t != -1.0f) { /* accepted: t <= closest, or t is NaN */
        if (!(t == t)) flags |= GFL_REF;
        const bool tie = (t == closest) && (best != GBEST_NONE);
        if (!tie || gen_rank(ranks, n_spheres, e) > gen_rank(ranks, n_spheres, best)) best = e;
        closest = t;
    }
}

/* the same test with the primitive's record already in registers (the leaf loops load the next record while they test this one) */
DEV void gen_leaf_test_rec(const DScene &sc, const int *chains, const uint32_t *ranks, int n_spheres, const DSphere &sp, const DQuad &qd, uint32_t e,
                           const Ray &ray, float ray_a, float &closest, uint32_t &best, int &flags) {
    const uint32_t cid = GENT_CHAIN(e);
    Ray r = ray;
    float ra = ray_a;
    if (cid != 0) {
        r = apply_chain(sc, ray, chains[2 * cid], chains[2 * cid + 1]);
        ra = vlen2(r.d);
    }
    float t;
    if (GENT_QUAD(e)) {
        float al, be;
        if (!quad_hit_t(qd, r, 0.001f, closest, t, al, be)) t = -1.0f;
    } else {
        t = gen_sphere_root(sp, r, ra, 0.001f, closest);
    }
    if (t != -1.0f) {
        if (!(t == t)) flags |= GFL_REF;
        const bool tie = (t == closest) && (best != GBEST_NONE);
        if (!tie || gen_rank(ranks, n_spheres, e) > gen_rank(ranks, n_spheres, best)) best = e;
        closest = t;
    }
}

/* The two halves of gen_leaf_test_rec, for the leaf loops that test TWO primitives per trip: the own roots of both are computed first, against the
 * same closest_so_far (two independent instruction chains for a wave that is bound by the latency of one), then applied in entry order.  That is the
 * sequential result: a primitive offers its near root if that lies in [t_min, t_max], else its far root; with the larger t_max a root in (new closest, old closest]
 * may be offered where the sequential test offers none -- gen_apply_t refuses it, as the sequential range test would have -- and a root <= the new closest is
 * offered by both.  NaN roots are accepted by both forms (every comparison with NaN is false) and flag the ray for the scan. */
DEV float gen_prim_t(const DScene &sc, const int *chains, const DSphere &sp, const DQuad &qd, uint32_t e, const Ray &ray, float ray_a, float closest) {
    const uint32_t cid = GENT_CHAIN(e);
    Ray r = ray;
    float ra = ray_a;
    if (cid != 0) {
        r = apply_chain(sc, ray, chains[2 * cid], chains[2 * cid + 1]);
        ra = vlen2(r.d);
    }
    float t;
    if (GENT_QUAD(e)) {
        float al, be;
        if (!quad_hit_t(qd, r, 0.001f, closest, t, al, be)) t = -1.0f;
    } else {
        t = gen_sphere_root(sp, r, ra, 0.001f, closest);
    }
    return t;
}
DEV void gen_apply_t(const uint32_t *ranks, int n_spheres, float t, uint32_t e, float &closest, uint32_t &best, int &flags) {
    if (t != -1.0f && !(t > closest)) { /* accepted: t <= closest_so_far, or t is NaN */
        if (!(t == t)) flags |= GFL_REF;
        const bool tie = (t == closest) && (best != GBEST_NONE);
        if (!tie || gen_rank(ranks, n_spheres, e) > gen_rank(ranks, n_spheres, best)) best = e;
        closest = t;
    }
}

/* world::hit's scan over the solids as the reference runs it (world.cuh:122-168 through the flattened items): the exact
 * answer for the rays the tree walk does not decide */
DEV void gen_scan_solids(const DScene &sc, int first_medium, const int *chains, int n_chains, const Ray &r, float &closest, uint32_t &best) {
    closest = __builtin_inff();
    Best b; b.kind = HIT_NONE; b.t = 0; b.prim = 0; b.chain_first = 0; b.chain_count = 0;
    for (int i = 0; i < first_medium; i++) {
        const DItem it = sc.items[i];
        if (it.kind == ITEM_SPHERES) run_spheres(sc, r, it.first, it.count, it.chain_first, it.chain_count, 0.001f, closest, b);
        else if (it.kind == ITEM_QUADS) run_quads(sc, r, it.first, it.count, it.chain_first, it.chain_count, 0.001f, closest, b);
    }
    best = GBEST_NONE;
    if (b.kind != HIT_NONE) {
        uint32_t cid = 0;
        if (b.chain_count > 0)
            for (int k = 1; k < n_chains; k++) if (chains[2 * k] == b.chain_first && chains[2 * k + 1] == b.chain_count) cid = (uint32_t)k;
        best = GENT(b.kind == HIT_QUAD ? 1u : 0u, cid, (uint32_t)b.prim);
    }
}

/* the constant media, after every solid, in scan order, with the solids' closest_so_far
 * (constant_medium::hit, objects.cuh:396-434; world.cuh:154-160) */
DEV void gen_media(const DScene &sc, int first_medium, int n_items, const Ray &ray, Rng &rng, float &closest, uint32_t &best) {
    for (int i = first_medium; i < n_items; i++) {
        const DItem it = sc.items[i];
        if (it.kind != ITEM_MEDIUM) continue;
        const Ray rm = apply_chain(sc, ray, it.chain_first, it.chain_count);
        float t1, t2;
        if (!boundary_t(sc, ray, it.first, it.count, -__builtin_inff(), __builtin_inff(), t1)) continue;
        if (!boundary_t(sc, ray, it.first, it.count, (float)((double)t1 + 0.0001), __builtin_inff(), t2)) continue;
        if (t1 < 0.001f) t1 = 0.001f;
        if (t2 > closest) t2 = closest;
        if (t1 >= t2) continue;
        if (t1 < 0) t1 = 0;
        const float ray_length = vlen(rm.d);
        const float distance_inside_boundary = (t2 - t1) * ray_length;
        const double hit_distance = sc.neg_inv_density[it.medium] * (double)mort_logf(random_float(rng));
        if (hit_distance > (double)distance_inside_boundary) continue;
        closest = (float)((double)t1 + hit_distance / (double)ray_length);
        best = GENT(0, GCHAIN_MEDIUM, (uint32_t)i);
    }
}

/* entry code of the winner -> what resolve_hit / shade_hit read */
DEV Best gen_decode_best(const DScene &sc, const int *chains, uint32_t best, float closest) {
    Best b;
    b.t = closest;
    const uint32_t cid = GENT_CHAIN(best);
    b.prim = (int)GENT_IDX(best);
    if (cid == GCHAIN_MEDIUM) {
        b.kind = HIT_MEDIUM;
        const DItem it = sc.items[b.prim];
        b.chain_first = it.chain_first; b.chain_count = it.chain_count;
    } else {
        b.kind = GENT_QUAD(best) ? HIT_QUAD : HIT_SPHERE;
        b.chain_first = chains[2 * cid]; b.chain_count = chains[2 * cid + 1];
    }
    return b;
}

/* ---- the whole closest-hit search as ONE lane runs it (host loop, CPU tests): the same steps the kernels schedule
 * across a wave.  Returns false on a miss.  scans: counts the segments the scan decided. ---- */
DEV bool gen_world_hit(const DScene &sc, const GenWalk &gw, const Ray &ray, Rng &rng, Best &out, unsigned long long *scans) {
    GenRay gr;
    const bool ordinary = gen_ray_setup(ray, gw.gx, gw.gy, gw.gz, gw.gR, gw.mnear, gw.kmin, gr);
    const float ray_a = vlen2(ray.d);
    float closest = __builtin_inff();
    uint32_t best = GBEST_NONE;
    int flags = ordinary ? 0 : GFL_REF;
    if (ordinary && gw.root != 0xffffu) {
        unsigned short stack[MORT_OWN_STACK];
        int sp = 0;
        uint32_t cur = gw.root;
        for (;;) {
            if (cur & 0x8000u) {
                uint32_t pos = GLEAF_FIRST(cur);
                for (int cnt = (int)GLEAF_COUNT(cur); cnt > 0; cnt--, pos++)
                    gen_leaf_test(sc, gw.chains, gw.ranks, gw.n_spheres, sc.spheres, sc.quads, gw.entries[pos], ray, ray_a, closest, best, flags);
                if (sp == 0) break;
                cur = stack[--sp];
                continue;
            }
            const DNodeQ nq = gw.nodes[cur & 0x7fffu];
            const GenBoxes nd = gen_node_decode(nq.ox, nq.oy, nq.oz, nq.exps, nq.q0, nq.q1, nq.q2, nq.children);
            float te0, te1;
            const bool m0 = gen_prune(nd.x0min, nd.x0max, nd.y0min, nd.y0max, nd.z0min, nd.z0max, gr, closest, te0);
            const bool m1 = gen_prune(nd.x1min, nd.x1max, nd.y1min, nd.y1max, nd.z1min, nd.z1max, gr, closest, te1);
            if (m0 && m1) {
                if (sp == 0) break;
                cur = stack[--sp];
            } else if (!m0 && !m1) {
                const bool first0 = te0 <= te1;
                stack[sp++] = (unsigned short)(first0 ? nd.c1 : nd.c0);
                cur = first0 ? nd.c0 : nd.c1;
            } else cur = m0 ? nd.c1 : nd.c0;
        }
    }
    if (flags) {
#if defined(MORT_DEBUG_SCANS) && !defined(__HIP_DEVICE_COMPILE__)
        printf("scan: flags %d best %08x closest %.9g o (%g %g %g) d (%g %g %g) inv (%g %g %g)\n", flags, best, closest, ray.o.x, ray.o.y, ray.o.z, ray.d.x, ray.d.y, ray.d.z, gr.ix, gr.iy, gr.iz);
#endif
        if (scans) *scans += 1;
        gen_scan_solids(sc, gw.first_medium, gw.chains, gw.n_chains, ray, closest, best);
    }
    gen_media(sc, gw.first_medium, sc.n_items, ray, rng, closest, best);
    if (best == GBEST_NONE) return false;
    out = gen_decode_best(sc, gw.chains, best, closest);
    return true;
}

#endif
