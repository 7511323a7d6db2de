/*
 * dev_shade_call.h -- dev_shade.h's shade_hit() as an out-of-line device function with everything passed in registers,
 * for the kernels that find the closest hit through the unified tree (mega_gen.hip, wave_gen.hip).
 */
#ifndef MORT_DEV_SHADE_CALL_H
#define MORT_DEV_SHADE_CALL_H

#include "dev_shade.h"

/* Shading is out of line: one call per segment with everything passed in registers.  Inlined into the state loop, hipcc 7.2 -O3
 * produced a kernel whose scattered-ray origin was wrong for a few rays per thousand.  Pass bisection (-mllvm -opt-bisect-limit,
 * scripts/bisect_build.sh + bisect_check.py, on wf_shade_gen) puts the first bad pass at SLPVectorizerPass on the kernel; it is not the
 * packed-fp32 lowering (still wrong with -packed-fp32-ops off) and not type-based aliasing (still wrong with -fno-strict-aliasing).
 * The build now carries -fno-slp-vectorize, with which the inlined form (-DMORT_SHADE_INLINE) is bit-exact on every GPU test -- and
 * no faster (final scene 800x800x100: 439 vs 442 ms), so the call stays: it keeps the shade step's registers out of the traversal
 * steps (186 vs 193 VGPRs).  The parity tests against the oracle are what guards this. */
struct ShadeRet { float ox, oy, oz, dx, dy, dz, tm, kx, ky, kz, rp, fx, fy, fz; int flags; uint32_t d, v0, v1, v2, v3, v4, draws; };
static __device__ __attribute__((noinline)) ShadeRet shade_call(const DScene *scp, int light_type, int light_idx, float ox, float oy, float oz, float dx, float dy, float dz, float tm,
                                                         float time0, float t, int kind, int prim, int cf, int cc, uint32_t d, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, uint32_t v4, uint32_t draws) {
    Ray ray; ray.o = mk(ox, oy, oz); ray.d = mk(dx, dy, dz); ray.tm = tm;
    Best b; b.t = t; b.kind = kind; b.prim = prim; b.chain_first = cf; b.chain_count = cc;
    Rng rng; rng.d = d; rng.v0 = v0; rng.v1 = v1; rng.v2 = v2; rng.v3 = v3; rng.v4 = v4; rng.draws = draws;
    const ShadeOut so = shade_hit(*scp, light_type, light_idx, ray, time0, b, rng);
    ShadeRet r;
    r.ox = ray.o.x; r.oy = ray.o.y; r.oz = ray.o.z; r.dx = ray.d.x; r.dy = ray.d.y; r.dz = ray.d.z; r.tm = ray.tm;
    r.kx = so.e.kx; r.ky = so.e.ky; r.kz = so.e.kz; r.rp = so.e.rp; r.fx = so.final_value.x; r.fy = so.final_value.y; r.fz = so.final_value.z;
    r.flags = (so.done ? 1 : 0) | (so.ident ? 2 : 0);
    r.d = rng.d; r.v0 = rng.v0; r.v1 = rng.v1; r.v2 = rng.v2; r.v3 = rng.v3; r.v4 = rng.v4; r.draws = rng.draws;
    return r;
}


/* the same call with the structs the callers hold */
DEV ShadeOut shade_hit_outlined(const DScene *scp, int light_type, int light_idx, Ray &ray, float time0, const Best &b, Rng &rng) {
    const ShadeRet r = shade_call(scp, light_type, light_idx, ray.o.x, ray.o.y, ray.o.z, ray.d.x, ray.d.y, ray.d.z, ray.tm, time0,
                                  b.t, b.kind, b.prim, b.chain_first, b.chain_count, rng.d, rng.v0, rng.v1, rng.v2, rng.v3, rng.v4, rng.draws);
    ShadeOut so;
    ray.o = mk(r.ox, r.oy, r.oz); ray.d = mk(r.dx, r.dy, r.dz); ray.tm = r.tm;
    so.e.kx = r.kx; so.e.ky = r.ky; so.e.kz = r.kz; so.e.rp = r.rp; so.final_value = mk(r.fx, r.fy, r.fz);
    so.done = r.flags & 1; so.ident = (r.flags & 2) != 0;
    rng.d = r.d; rng.v0 = r.v0; rng.v1 = r.v1; rng.v2 = r.v2; rng.v3 = r.v3; rng.v4 = r.v4; rng.draws = r.draws;
    return so;
}

#endif
