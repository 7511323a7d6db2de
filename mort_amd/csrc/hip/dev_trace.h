/*
 * dev_trace.h -- closest-hit search, hit-record reconstruction, textures,
 * material scatter and light sampling for the gfx950 kernels.
 *
 * Structure differs from the reference on purpose:
 *   - world::hit's seven dispatch loops + recursion (world.cuh:104-171,
 *     objects.cuh:858-887) run as one loop over flattened ITEMS;
 *   - the closest-hit search only tracks (t, primitive); the hit record
 *     (point, normal, front_face, uv -- objects.cuh:79-85,206-212) is rebuilt
 *     once for the winner, so acos/atan2 run once per segment, not once per
 *     accepted candidate.  Every rebuilt field is a pure function of (ray, t,
 *     primitive), so it is bit-identical to computing it at hit time;
 *   - aabb::hit's per-node `1.0 / dir[a]` in fp64 (aabb.cuh:40) is hoisted to
 *     once per ray;
 *   - the BVH walk is threaded (no stack); order and t_max flow are those of
 *     objects.cuh:664-723 (left first, right child sees the left's result).
 * Results are bit-identical to the oracle's recursive restatement.
 */
#ifndef MORT_DEV_TRACE_H
#define MORT_DEV_TRACE_H

#include "dev_math.h"
#include "dev_scene.h"
#include "mort_scene.h"

enum { HIT_NONE = 0, HIT_SPHERE = 1, HIT_QUAD = 2, HIT_MEDIUM = 3 };

struct Best {
    float t;
    int kind;
    int prim;        /* sphere/quad: index in scene arrays; medium: item index */
    int chain_first, chain_count;
};

struct HitRec { /* hit_record.cuh:10-19 */
    V3 p, normal;
    uint32_t mat;
    float t, u, v;
    bool front_face;
    /* a sphere's (u, v) (objects.cuh:101-108: acos + atan2) is a pure function of its outward normal and only the image texture and the
     * error pattern read it: resolve_hit keeps the normal (uv_sphere) and texture_value_rec computes u, v where they are read */
    V3 on;
    bool uv_sphere;
};

/* ---- transform chains (translate::hit / rotate_y::hit, objects.cuh:268-278,334-366) ---- */
DEV Ray apply_chain(const DScene &sc, Ray r, int first, int count) {
    for (int k = 0; k < count; k++) {
        const DXform x = sc.xforms[first + k];
        if (x.kind == XF_TRANSLATE) {
            r.o = vsub(r.o, mk(x.a, x.b, x.c));
        } else {
            const float st = x.a, ct = x.b;
            V3 o = r.o, d = r.d;
            o.x = ct * r.o.x - st * r.o.z;
            o.z = st * r.o.x + ct * r.o.z;
            d.x = ct * r.d.x - st * r.d.z;
            d.z = st * r.d.x + ct * r.d.z;
            r.o = o; r.d = d;
        }
    }
    return r;
}
DEV void unapply_chain(const DScene &sc, int first, int count, V3 &p, V3 &n) {
    for (int k = count - 1; k >= 0; k--) {
        const DXform x = sc.xforms[first + k];
        if (x.kind == XF_TRANSLATE) {
            p = vadd(p, mk(x.a, x.b, x.c));
        } else {
            const float st = x.a, ct = x.b;
            V3 q = p, m = n;
            q.x = ct * p.x + st * p.z;
            q.z = -st * p.x + ct * p.z;
            m.x = ct * n.x + st * n.z;
            m.z = -st * n.x + ct * n.z;
            p = q; n = m;
        }
    }
}

/* ---- primitive tests returning only t (objects.cuh:60-77,190-204) ---- */
DEV V3 sphere_center(const DSphere &s, float time) { /* objects.cuh:90-97 */
    V3 c = mk(s.cx, s.cy, s.cz);
    if (!(s.mat >> 31)) return c;
    return vadd(c, vscale(time, mk(s.vx, s.vy, s.vz)));
}
DEV bool sphere_hit_t(const DSphere &s, const Ray &r, float a, float t_min, float t_max, float &t_out) {
    const V3 oc = vsub(r.o, sphere_center(s, r.tm));
    const float half_b = vdot(oc, r.d);
    const float c = vlen2(oc) - s.radius * s.radius;
    const float discriminant = half_b * half_b - a * c;
    if (discriminant < 0) return false;
    const float sqrtd = mort_sqrtf(discriminant);
    float root = (-half_b - sqrtd) / a;
    if (root < t_min || t_max < root) {
        root = (-half_b + sqrtd) / a;
        if (root < t_min || t_max < root) return false;
    }
    t_out = root;
    return true;
}
DEV bool quad_hit_t(const DQuad &q, const Ray &r, float t_min, float t_max, float &t_out, float &alpha_out, float &beta_out) {
    const V3 n = mk(q.n[0], q.n[1], q.n[2]);
    float denom = vdot(n, r.d);
    if ((double)mort_fabsf(denom) < 1e-8) return false;
    float t = (q.D - vdot(n, r.o)) / denom;
    if (t < t_min || t > t_max) return false;
    V3 intersection = ray_at(r, t);
    V3 planar = vsub(intersection, mk(q.Q[0], q.Q[1], q.Q[2]));
    const V3 w = mk(q.w[0], q.w[1], q.w[2]);
    float alpha = vdot(w, vcross(planar, mk(q.v[0], q.v[1], q.v[2])));
    float beta = vdot(w, vcross(mk(q.u[0], q.u[1], q.u[2]), planar));
    if ((alpha < 0) || (alpha > 1) || (beta < 0) || (beta > 1)) return false;
    t_out = t; alpha_out = alpha; beta_out = beta;
    return true;
}

/* ---- aabb::hit with the fp64 reciprocal hoisted (aabb.cuh:37-59) ---- */
struct InvDir { double x, y, z; };
DEV InvDir inv_dir(const Ray &r) { InvDir i; i.x = 1.0 / (double)r.d.x; i.y = 1.0 / (double)r.d.y; i.z = 1.0 / (double)r.d.z; return i; }

#define SLAB(IMIN, IMAX, ORIG, INV)                                   \
    {                                                                 \
        double t0 = (double)((IMIN) - (ORIG)) * (INV);                \
        double t1 = (double)((IMAX) - (ORIG)) * (INV);                \
        if ((INV) < 0) { double aux = t0; t0 = t1; t1 = aux; }        \
        if (t0 > (double)t_min) t_min = (float)t0;                    \
        if (t1 < (double)t_max) t_max = (float)t1;                    \
        if (t_max <= t_min) return false;                             \
    }
DEV bool aabb_hit(const DBvhNode &b, const Ray &r, const InvDir &inv, float t_min, float t_max) {
    SLAB(b.xmin, b.xmax, r.o.x, inv.x)
    SLAB(b.ymin, b.ymax, r.o.y, inv.y)
    SLAB(b.zmin, b.zmax, r.o.z, inv.z)
    return true;
}
#undef SLAB

/* ---- aabb::hit as a BOOLEAN, decided in fp32 whenever that is provably the reference's answer ----
 *
 * The reference computes, per plane, a = fl32(plane - orig), t64 = fl64(a * inv64) with
 * inv64 = fl64(1.0 / dir), narrows t64 to fp32 where it updates t_min / t_max, and reports a miss iff
 * t_max <= t_min after the three axes (aabb.cuh:37-59; equivalently r = fl32(t64) per plane,
 * t_min = max(0.001, min(r0,r1) per axis), t_max = min(closest, max(r0,r1) per axis): see mega_bvh.h).
 * Only that boolean is used.  Let inv32 = fl32(inv64) and p = fl32(a * inv32).  For finite normal values
 * p / r = (1+e1)(1+e2) / ((1+e3)(1+e4)) with |e1|,|e2|,|e4| <= 2^-24, |e3| <= 2^-53, so |p - r| <= 2^-22 |p|.
 * max / min are 1-Lipschitz, so |gap_p - gap_r| <= 2 * 2^-22 * M with M = max |p| over the six planes, where
 * gap = t_max - t_min; the fp32 subtraction adds <= 2^-24 * 2M.  With tau = 2^-20 * M:
 *     gap_p >  tau  =>  gap_r > 0  (reference: hit)        gap_p < -tau  =>  gap_r < 0  (reference: miss)
 * and only |gap_p| <= tau (about one box test in 10^5) needs the exact fp64 evaluation.  Rays whose
 * reciprocal is not a normal finite fp32 (zero / denormal / huge direction components: inf, NaN products)
 * always take the exact path (`exact_only`).  The result is the reference's boolean in every case.
 */
struct SlabRay {
    float ox, oy, oz;
    float i32x, i32y, i32z; /* fl32(inv64) */
    double ix, iy, iz;      /* inv64 = 1.0 / (double)dir */
    bool exact_only;
};
DEV bool slab_inv_ok(float v) { const float a = mort_fabsf(v); return a > 1e-30f && a < 1e30f; }
DEV SlabRay slab_ray(float ox, float oy, float oz, float dx, float dy, float dz) {
    SlabRay s;
    s.ox = ox; s.oy = oy; s.oz = oz;
    s.ix = 1.0 / (double)dx; s.iy = 1.0 / (double)dy; s.iz = 1.0 / (double)dz;
    s.i32x = (float)s.ix; s.i32y = (float)s.iy; s.i32z = (float)s.iz;
    s.exact_only = !(slab_inv_ok(s.i32x) && slab_inv_ok(s.i32y) && slab_inv_ok(s.i32z));
    return s;
}
DEV bool slab_exact(const DBvhNode &nd, float ox, float oy, float oz, double ix, double iy, double iz, float closest) {
    const bool nx = ix < 0, ny = iy < 0, nz = iz < 0;
    const float x0 = nx ? nd.xmax : nd.xmin, x1 = nx ? nd.xmin : nd.xmax;
    const float y0 = ny ? nd.ymax : nd.ymin, y1 = ny ? nd.ymin : nd.ymax;
    const float z0 = nz ? nd.zmax : nd.zmin, z1 = nz ? nd.zmin : nd.zmax;
    float t_min = 0.001f, t_max = closest;
    t_min = __builtin_fmaxf(t_min, (float)((double)(x0 - ox) * ix));
    t_max = __builtin_fminf(t_max, (float)((double)(x1 - ox) * ix));
    t_min = __builtin_fmaxf(t_min, (float)((double)(y0 - oy) * iy));
    t_max = __builtin_fminf(t_max, (float)((double)(y1 - oy) * iy));
    t_min = __builtin_fmaxf(t_min, (float)((double)(z0 - oz) * iz));
    t_max = __builtin_fminf(t_max, (float)((double)(z1 - oz) * iz));
    return !(t_max <= t_min);
}
/* returns true when the box is hit (reference: !(t_max <= t_min)) */
DEV bool slab_hit(const DBvhNode &nd, const SlabRay &r, float closest) {
    const float px0 = (nd.xmin - r.ox) * r.i32x, px1 = (nd.xmax - r.ox) * r.i32x;
    const float py0 = (nd.ymin - r.oy) * r.i32y, py1 = (nd.ymax - r.oy) * r.i32y;
    const float pz0 = (nd.zmin - r.oz) * r.i32z, pz1 = (nd.zmax - r.oz) * r.i32z;
    const float t_min = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(px0, px1), __builtin_fminf(py0, py1)),
                                        __builtin_fmaxf(__builtin_fminf(pz0, pz1), 0.001f));
    const float t_max = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(px0, px1), __builtin_fmaxf(py0, py1)),
                                        __builtin_fminf(__builtin_fmaxf(pz0, pz1), closest));
    const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(mort_fabsf(px0), mort_fabsf(px1)), __builtin_fmaxf(mort_fabsf(py0), mort_fabsf(py1))),
                                    __builtin_fmaxf(mort_fabsf(pz0), mort_fabsf(pz1)));
    const float gap = t_max - t_min;
    const float tau = m * 9.5367431640625e-07f; /* 2^-20 */
    /* decided only when everything is finite, away from the denormal range, and outside the error band */
    const bool decided = !r.exact_only && (mort_fabsf(gap) > tau) && (m < 1e30f) && (m > 1e-30f);
    bool hit = gap > 0;
    if (!decided) hit = slab_exact(nd, r.ox, r.oy, r.oz, r.ix, r.iy, r.iz, closest);
    return hit;
}

/* ---- closest hit over a run of primitives sharing one chain ---- */
DEV void run_spheres(const DScene &sc, const Ray &rw, int first, int count, int cf, int cc, float t_min, float &closest, Best &best) {
    const Ray r = apply_chain(sc, rw, cf, cc);
    const float a = vlen2(r.d);
    for (int i = first; i < first + count; i++) {
        const DSphere s = sc.spheres[i];
        float t;
        if (sphere_hit_t(s, r, a, t_min, closest, t)) {
            closest = t;
            best.t = t; best.kind = HIT_SPHERE; best.prim = i; best.chain_first = cf; best.chain_count = cc;
        }
    }
}
DEV void run_quads(const DScene &sc, const Ray &rw, int first, int count, int cf, int cc, float t_min, float &closest, Best &best) {
    const Ray r = apply_chain(sc, rw, cf, cc);
    for (int i = first; i < first + count; i++) {
        float t, al, be;
        if (quad_hit_t(sc.quads[i], r, t_min, closest, t, al, be)) {
            closest = t;
            best.t = t; best.kind = HIT_QUAD; best.prim = i; best.chain_first = cf; best.chain_count = cc;
        }
    }
}

/* ---- bvh::hit, threaded (objects.cuh:664-723) ---- */
DEV void run_bvh(const DScene &sc, const Ray &r, int first, int count, float t_min, float &closest, Best &best) {
    const SlabRay sr = slab_ray(r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z);
    const float a = vlen2(r.d);
    const int end = first + count;
    int node = first;
    while (node < end) {
        const DBvhNode nd = sc.nodes[node];
        const bool leaf = nd.skip >> 31;
        const int skip = (int)(nd.skip & 0x7fffffffu);
        if (!slab_hit(nd, sr, closest)) { node = skip; continue; } /* t_min = 0.001f: the only value world::hit passes */
        if (!leaf) { node = node + 1; continue; }
        const uint32_t pa = nd.prims & 0xffffu, pb = nd.prims >> 16;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const uint32_t p = k ? pb : pa;
            if (k == 1 && pb == pa) break; /* span-1 leaf stores its object twice (objects.cuh:555-563); the retest is idempotent */
            const int idx = (int)(p & 0x7fffu);
            if (p >> 15) {
                float t, al, be;
                if (quad_hit_t(sc.quads[idx], r, t_min, closest, t, al, be)) {
                    closest = t; best.t = t; best.kind = HIT_QUAD; best.prim = idx; best.chain_first = 0; best.chain_count = 0;
                }
            } else {
                float t;
                if (sphere_hit_t(sc.spheres[idx], r, a, t_min, closest, t)) {
                    closest = t; best.t = t; best.kind = HIT_SPHERE; best.prim = idx; best.chain_first = 0; best.chain_count = 0;
                }
            }
        }
        node = skip;
    }
}

/* closest t over a medium's boundary sub-items in [t_min, t_max] (hitDispatch of
 * constant_medium::hit, objects.cuh:400-406) */
DEV bool boundary_t(const DScene &sc, const Ray &rw, int first, int count, float t_min, float t_max, float &t_out) {
    Best b; b.kind = HIT_NONE; b.t = 0; b.prim = 0; b.chain_first = 0; b.chain_count = 0;
    float closest = t_max;
    for (int i = first; i < first + count; i++) {
        const DItem it = sc.subitems[i];
        if (it.kind == ITEM_SPHERES) run_spheres(sc, rw, it.first, it.count, it.chain_first, it.chain_count, t_min, closest, b);
        else if (it.kind == ITEM_QUADS) run_quads(sc, rw, it.first, it.count, it.chain_first, it.chain_count, t_min, closest, b);
    }
    t_out = b.t;
    return b.kind != HIT_NONE;
}

/* ---- world::hit (world.cuh:104-171) ---- */
DEV bool world_hit(const DScene &sc, const Ray &r, Rng &rng, Best &best) {
    const float t_min = 0.001f;
    float closest = __builtin_inff();
    best.kind = HIT_NONE; best.t = 0; best.prim = 0; best.chain_first = 0; best.chain_count = 0;
    for (int i = 0; i < sc.n_items; i++) {
        const DItem it = sc.items[i];
        /* sequential ifs on the (wave-uniform) item kind, not an if / else-if chain: each merges a modified (closest, best, stream) with the unmodified one,
         * so hipcc updates the registers in place instead of copying them around a common join (mega_bvh.h) */
        const int kind = it.kind;
        if (kind == ITEM_BVH) run_bvh(sc, r, it.first, it.count, t_min, closest, best);
        if (kind == ITEM_SPHERES) run_spheres(sc, r, it.first, it.count, it.chain_first, it.chain_count, t_min, closest, best);
        if (kind == ITEM_QUADS) run_quads(sc, r, it.first, it.count, it.chain_first, it.chain_count, t_min, closest, best);
        if (kind != ITEM_BVH && kind != ITEM_SPHERES && kind != ITEM_QUADS) { /* constant_medium::hit, objects.cuh:396-434 */
            const Ray rm = apply_chain(sc, r, it.chain_first, it.chain_count);
            float t1, t2;
            if (!boundary_t(sc, r, it.first, it.count, -__builtin_inff(), __builtin_inff(), t1)) continue;
            if (!boundary_t(sc, r, it.first, it.count, (float)((double)t1 + 0.0001), __builtin_inff(), t2)) continue;
            if (t1 < t_min) t1 = t_min;
            if (t2 > closest) t2 = closest;
            if (t1 >= t2) continue;
            if (t1 < 0) t1 = 0;
            const float ray_length = vlen(rm.d);
            const float distance_inside_boundary = (t2 - t1) * ray_length;
            const double hit_distance = sc.neg_inv_density[it.medium] * (double)mort_logf(random_float(rng));
            if (hit_distance > (double)distance_inside_boundary) continue;
            const float t = (float)((double)t1 + hit_distance / (double)ray_length);
            closest = t;
            best.t = t; best.kind = HIT_MEDIUM; best.prim = i; best.chain_first = it.chain_first; best.chain_count = it.chain_count;
        }
    }
    return best.kind != HIT_NONE;
}

/* ---- rebuild the winner's hit record ---- */
DEV void sphere_uv(V3 p, float &u, float &v) { /* objects.cuh:101-108 */
    float theta = mort_acosf(-p.y);
    float phi = (float)((double)mort_atan2f(-p.z, p.x) + 3.141592565);
    u = (float)((double)phi / (2.0 * 3.141592565));
    v = (float)((double)theta / 3.141592565);
}
DEV void resolve_hit(const DScene &sc, const Ray &rw, const Best &b, HitRec &rec) {
    const Ray r = apply_chain(sc, rw, b.chain_first, b.chain_count);
    rec.on = mk(0, 0, 0); rec.uv_sphere = false;
    rec.t = b.t;
    rec.p = ray_at(r, b.t);
    if (b.kind == HIT_SPHERE) {
        const DSphere s = sc.spheres[b.prim];
        const V3 outward = vdiv(vsub(rec.p, sphere_center(s, r.tm)), s.radius);
        rec.front_face = vdot(r.d, outward) < 0;
        rec.normal = rec.front_face ? outward : vneg(outward);
        rec.on = outward; rec.uv_sphere = true; rec.u = 0; rec.v = 0;
        rec.mat = s.mat & 0x7fffffffu;
    } else if (b.kind == HIT_QUAD) {
        const DQuad q = sc.quads[b.prim];
        const V3 planar = vsub(rec.p, mk(q.Q[0], q.Q[1], q.Q[2]));
        const V3 w = mk(q.w[0], q.w[1], q.w[2]);
        rec.u = vdot(w, vcross(planar, mk(q.v[0], q.v[1], q.v[2])));
        rec.v = vdot(w, vcross(mk(q.u[0], q.u[1], q.u[2]), planar));
        const V3 n = mk(q.n[0], q.n[1], q.n[2]);
        rec.front_face = vdot(r.d, n) < 0;
        rec.normal = rec.front_face ? n : vneg(n);
        rec.mat = q.mat;
    } else { /* medium: objects.cuh:425-431 */
        rec.normal = mk(1, 0, 0);
        rec.front_face = true;
        rec.u = 0; rec.v = 0;
        rec.mat = sc.items[b.prim].mat;
    }
    unapply_chain(sc, b.chain_first, b.chain_count, rec.p, rec.normal);
}

/* ---- textures (textures.cuh) ---- */
DEV float clamp01(float x) { if (x < 0) return 0; if (x > 1) return 1; return x; }

DEV float perlin_noise(const float *nt, V3 p) { /* textures.cuh:174-196,232-250 */
    const float *ranvec = nt;
    const int *perm_x = (const int *)(nt + 3 * MORT_POINT_COUNT);
    const int *perm_y = perm_x + MORT_POINT_COUNT;
    const int *perm_z = perm_y + MORT_POINT_COUNT;
    float fu = p.x - mort_floorf(p.x);
    float fv = p.y - mort_floorf(p.y);
    float fw = p.z - mort_floorf(p.z);
    fu = fu * fu * (3 - 2 * fu);
    fv = fv * fv * (3 - 2 * fv);
    fw = fw * fw * (3 - 2 * fw);
    const int i = mort_f2i(mort_floorf(p.x));
    const int j = mort_f2i(mort_floorf(p.y));
    const int k = mort_f2i(mort_floorf(p.z));
    const double u = fu, v = fv, w = fw;
    const double uu = u * u * (3 - 2 * u);
    const double vv = v * v * (3 - 2 * v);
    const double ww = w * w * (3 - 2 * w);
    double accum = 0.0;
    for (int di = 0; di < 2; di++)
        for (int dj = 0; dj < 2; dj++)
            for (int dk = 0; dk < 2; dk++) {
                const int idx = perm_x[(i + di) & 255] ^ perm_y[(j + dj) & 255] ^ perm_z[(k + dk) & 255];
                const V3 c = mk(ranvec[3 * idx], ranvec[3 * idx + 1], ranvec[3 * idx + 2]);
                const V3 weight_v = mk((float)(u - di), (float)(v - dj), (float)(w - dk));
                accum += (di * uu + (1 - di) * (1 - uu)) * (dj * vv + (1 - dj) * (1 - vv)) *
                         (dk * ww + (1 - dk) * (1 - ww)) * (double)vdot(c, weight_v);
            }
    return (float)accum;
}
/* out of line: 7 x 8 gradient fetches, fp64 accumulators (textures.cuh:198-202,252-265).  Takes the table pointer, not
 * the scene: a kernel's scene view with LDS-resident tables then never escapes to a call */
static __host__ __device__ __attribute__((noinline)) V3 noise_value(const float *noise, int idx, V3 p) {
    const float *nt = noise + (size_t)idx * (sizeof(mort_noise_texture) / 4);
    const float scale = nt[3 * MORT_POINT_COUNT + 3 * MORT_POINT_COUNT];
    const V3 s = vscale(scale, p);
    double accum = 0.0;
    V3 temp_p = s;
    double weight = 1.0;
    for (int i = 0; i < 7; i++) {
        accum += weight * (double)perlin_noise(nt, temp_p);
        weight *= 0.5;
        temp_p = vscale(2, temp_p);
    }
    const float turb = mort_fabsf((float)accum);
    const V3 half = vscale(0.5f, mk(1, 1, 1));
    return vscale((float)(1 + mort_sin((double)s.z + 10.0 * (double)turb)), half);
}
static __host__ __device__ __attribute__((noinline)) V3 image_value(const DImage *image, const unsigned char *texels, int idx, float u, float v) { /* textures.cuh:129-146 */
    const DImage im = image[idx];
    if (im.height <= 0) return mk(0, 1, 1);
    u = clamp01(u);
    v = (float)(1.0 - (double)clamp01(v));
    int i = mort_f2i(u * (float)im.width);
    int j = mort_f2i(v * (float)im.height);
    const int row_bytes = im.width * 3;
    if (j < 0) j = 0;
    if (j > im.height - 1) j = im.height - 1;
    int rgb[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        int x = i * 3 + k;
        if (x < 0) x = 0;
        if (x > row_bytes - 1) x = row_bytes - 1;
        rgb[k] = texels[(size_t)im.offset + (size_t)j * row_bytes + x];
    }
    const float color_scale = (float)(1.0 / 255.0);
    return mk(color_scale * (float)rgb[0], color_scale * (float)rgb[1], color_scale * (float)rgb[2]);
}
DEV V3 texture_value(const DScene &sc, uint32_t tex, float u, float v, V3 p) { /* textures.cuh:327-349 */
    /* a checker's children may be checkers again: resolve iteratively */
    for (int guard = 0; guard < 8; guard++) {
        const int type = DREF_TYPE(tex), idx = DREF_IDX(tex);
        if (type == MORT_TEXTURE_SOLID) { const DSolid s = sc.solid[idx]; return mk(s.r, s.g, s.b); }
        if (type == MORT_TEXTURE_CHECKER) { /* textures.cuh:52-60 */
            const DChecker c = sc.checker[idx];
            const int xi = mort_f2i(mort_floorf(c.inv_scale * p.x));
            const int yi = mort_f2i(mort_floorf(c.inv_scale * p.y));
            const int zi = mort_f2i(mort_floorf(c.inv_scale * p.z));
            tex = ((xi + yi + zi) % 2 == 0) ? c.even : c.odd;
            continue;
        }
        if (type == MORT_TEXTURE_IMAGE) return image_value(sc.image_hbm, sc.texels, idx, u, v);
        if (type == MORT_TEXTURE_NOISE) return noise_value(sc.noise, idx, p);
        break;
    }
    const float error = (float)((mort_f2i(mort_floorf((float)((double)u * 1000.0))) % 2) ==
                                (mort_f2i(mort_floorf((float)((double)v * 1000.0))) % 2));
    return mk(error, 0.0f, error);
}
DEV V3 lambert_color(const DScene &sc, const DLambert &m, float u, float v, V3 p) {
    if (m.tex == 0) return mk(m.r, m.g, m.b);
    return texture_value(sc, m.tex, u, v, p);
}
/* texture_value for a resolved hit: the same lookups, a sphere's (u, v) computed only by the texture kinds that read them */
DEV V3 texture_value_rec(const DScene &sc, uint32_t tex, const HitRec &rec) {
    for (int guard = 0; guard < 8; guard++) {
        const int type = DREF_TYPE(tex), idx = DREF_IDX(tex);
        if (type == MORT_TEXTURE_SOLID) { const DSolid s = sc.solid[idx]; return mk(s.r, s.g, s.b); }
        if (type == MORT_TEXTURE_CHECKER) { /* textures.cuh:52-60 */
            const DChecker c = sc.checker[idx];
            const int xi = mort_f2i(mort_floorf(c.inv_scale * rec.p.x));
            const int yi = mort_f2i(mort_floorf(c.inv_scale * rec.p.y));
            const int zi = mort_f2i(mort_floorf(c.inv_scale * rec.p.z));
            tex = ((xi + yi + zi) % 2 == 0) ? c.even : c.odd;
            continue;
        }
        if (type == MORT_TEXTURE_NOISE) return noise_value(sc.noise, idx, rec.p);
        break; /* image texture, or an unknown tag's error pattern: these read (u, v) */
    }
    float u = rec.u, v = rec.v;
    if (rec.uv_sphere) sphere_uv(rec.on, u, v);
    return texture_value(sc, tex, u, v, rec.p);
}
DEV V3 lambert_color_rec(const DScene &sc, const DLambert &m, const HitRec &rec) {
    if (m.tex == 0) return mk(m.r, m.g, m.b);
    return texture_value_rec(sc, m.tex, rec);
}

/* ---- light sampling (objects.cuh:110-145,217-235,488-504; pdf.cuh:60-80) ---- */
struct Light { int kind; int idx; }; /* LIGHT_* ; idx = world sphere/quad/list index */

DEV float wsphere_pdf_value(const DSphere &s, V3 origin, V3 direction) {
    Ray r; r.o = origin; r.d = direction; r.tm = 0;
    float t;
    if (!sphere_hit_t(s, r, vlen2(direction), 0.001f, __builtin_huge_valf(), t)) return 0.0f;
    const V3 c1 = mk(s.cx, s.cy, s.cz);
    const float cos_theta_max = mort_sqrtf(1 - s.radius * s.radius / vlen2(vsub(c1, origin)));
    const float solid_angle = (float)(2 * 3.1415926 * (double)(1 - cos_theta_max));
    return (float)(1.0 / (double)solid_angle);
}
DEV V3 wsphere_random(const DSphere &s, V3 origin, Rng &rng) {
    const V3 direction = vsub(mk(s.cx, s.cy, s.cz), origin);
    const float distance_squared = vlen2(direction);
    const Onb uvw = onb_from_w(direction);
    const float r1 = random_float(rng);
    const float r2 = random_float(rng);
    const float z = 1 + r2 * (mort_sqrtf(1 - s.radius * s.radius / distance_squared) - 1);
    const float phi = (float)(2 * 3.141592 * (double)r1);
    const float sq = mort_sqrtf(1 - z * z);
    const float x = mort_cosf(phi) * sq;
    const float y = mort_sinf(phi) * sq;
    return onb_local(uvw, mk(x, y, z));
}
DEV float wquad_pdf_value(const DQuad &q, V3 origin, V3 direction) {
    Ray r; r.o = origin; r.d = direction; r.tm = 0;
    float t, al, be;
    if (!quad_hit_t(q, r, 0.001f, __builtin_huge_valf(), t, al, be)) return 0;
    const V3 n = mk(q.n[0], q.n[1], q.n[2]);
    const V3 normal = (vdot(direction, n) < 0) ? n : vneg(n);
    const float distance_squared = t * t * vlen2(direction);
    const float cosine = mort_fabsf(vdot(direction, normal) / vlen(direction));
    return distance_squared / (cosine * q.area);
}
DEV V3 wquad_random(const DQuad &q, V3 origin, Rng &rng) {
    const float a = random_float(rng);
    const float b = random_float(rng);
    const V3 p = vadd(vadd(mk(q.Q[0], q.Q[1], q.Q[2]), vscale(a, mk(q.u[0], q.u[1], q.u[2]))), vscale(b, mk(q.v[0], q.v[1], q.v[2])));
    return vsub(p, origin);
}

#endif
