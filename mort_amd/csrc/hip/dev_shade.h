/*
 * dev_shade.h -- one iteration of ray_color's loop body after world::hit has found the closest hit
 * (camera.cuh:100-151): rebuild the hit record, evaluate emission / scatter of its material, choose the next
 * direction (material pdf, or the light / material mixture of pdf.cuh:83-107 when the camera names a light object)
 * and form the bounce-stack entry that the unwind (camera.cuh:165-173) will read.
 *
 * Shared by every kernel that shades arbitrary scene graphs: mega_kernel (one lane per pixel), mega_gen_kernel
 * (state machine over this build's unified tree), wf_shade_gen (wavefront mode) and the host loop of
 * mort_hip_render_host -- one body, so the four forms cannot drift apart.
 */
#ifndef MORT_DEV_SHADE_H
#define MORT_DEV_SHADE_H

#include "dev_render.h"

#pragma clang fp contract(off)

#if defined(MORT_SHADE_MARKS) && defined(__HIP_DEVICE_COMPILE__) /* comments in the ISA at the shade step's parts (static instruction counts) */
#define SHMARK(name) asm volatile("; MARK " name)
#else
#define SHMARK(name) do { } while (0)
#endif

struct ShadeOut {
    bool done;         /* path ends here: final_value is the radiance of this segment (emission, or 0) */
    bool ident;        /* scatter whose stack entry is the identity (dielectric: k = (1,1,1), 1/pdf = 1) */
    StackEntry e;      /* k = scattering_pdf * attenuation, rp = 1 / pdf (valid when !done && !ident) */
    V3 final_value;
};

/* `ray` is the ray world::hit was called with; on a scatter it is replaced by the scattered ray.
 * ray_time0 = time of the sample's camera ray (lambertian / isotropic scatter with r.time(), camera.cuh:119). */
DEV ShadeOut shade_hit(const DScene &sc, int light_type, int light_idx, Ray &ray, float ray_time0, const Best &best, Rng &rng) {
    ShadeOut o;
    o.done = false; o.ident = false;
    o.e.kx = o.e.ky = o.e.kz = o.e.rp = 1.0f;
    o.final_value = mk(0, 0, 0);
    HitRec rec;
    SHMARK("resolve");
    resolve_hit(sc, ray, best, rec);
    SHMARK("dispatch");
    const int mtype = DREF_TYPE(rec.mat), midx = DREF_IDX(rec.mat);
    if (mtype == MORT_MAT_METAL) { /* materials.cuh:73-84 */
        SHMARK("metal");
        const DMetal m = sc.metal[midx];
        V3 reflected = reflect(ray.d, rec.normal);
        reflected = vadd(vunit(reflected), vscale(m.fuzz, random_unit_vector(rng)));
        ray.o = rec.p; ray.d = reflected; /* time stays r_in.time() */
        o.e.kx = 1.0f * m.r; o.e.ky = 1.0f * m.g; o.e.kz = 1.0f * m.b; o.e.rp = 1 / 1.0f;
    } else if (mtype == MORT_MAT_DIELECTRIC) { /* materials.cuh:107-130 */
        SHMARK("dielectric");
        const DDielectric m = sc.dielectric[midx];
        const float refraction_ratio = rec.front_face ? m.inv_ior : m.ior;
        const V3 unit_direction = vunit(ray.d);
        const float cos_theta = (float)mort_fmin((double)vdot(vneg(unit_direction), rec.normal), 1.0);
        const float sin_theta = (float)mort_sqrt(1.0 - (double)(cos_theta * cos_theta));
        const bool cant_refract = (double)(refraction_ratio * sin_theta) > 1.0;
        V3 direction;
        if (cant_refract || reflectance(cos_theta, refraction_ratio) > random_float(rng))
            direction = reflect(unit_direction, rec.normal);
        else
            direction = refract(unit_direction, rec.normal, refraction_ratio);
        ray.o = rec.p; ray.d = direction;
        o.ident = true; /* entry (1,1,1), 1/pdf = 1 */
    } else if (mtype == MORT_MAT_LAMBERTIAN || mtype == MORT_MAT_ISOTROPIC) {
        /* materials.cuh:38-44,182-188 + camera.cuh:115-145 */
        SHMARK("diffuse_texture");
        const bool lamb = (mtype == MORT_MAT_LAMBERTIAN);
        const DLambert m = lamb ? sc.lambert[midx] : sc.isotropic[midx];
        const V3 attenuation = lambert_color_rec(sc, m, rec);
        SHMARK("diffuse_sample");
        Onb uvw;
        if (lamb) uvw = onb_from_w(rec.normal);
        V3 dir;
        bool from_light = false;
        if (light_type != -1) from_light = random_float(rng) < 0.5; /* mixture_pdf::generate, pdf.cuh:96-103 */
        if (from_light) dir = light_random(sc, light_type, light_idx, rec.p, rng);
        else if (lamb) dir = onb_local(uvw, random_cosine_direction(rng));
        else dir = random_unit_vector(rng);
        SHMARK("diffuse_pdf");
        /* srec.pdf_ptr->value(dir): cosine_pdf / sphere_pdf (pdf.cuh:29-32,45-49) */
        float mat_pdf;
        if (lamb) {
            const float cosine_theta = vdot(vunit(dir), uvw.w);
            mat_pdf = mort_fmaxf(0, (float)((double)cosine_theta / 3.1415926));
        } else {
            mat_pdf = (float)(1 / (4 * 3.1415926));
        }
        float pdf = mat_pdf;
        if (light_type != -1) /* mixture_pdf::value, pdf.cuh:91-93 */
            pdf = (float)(0.5 * (double)light_pdf_value(sc, light_type, light_idx, rec.p, dir) + 0.5 * (double)mat_pdf);
        float scattering_pdf; /* materials.cuh:51-55,195-198 */
        if (lamb) {
            const float cos_theta = vdot(rec.normal, vunit(dir));
            scattering_pdf = (cos_theta < 0) ? 0.0f : (float)((double)cos_theta / 3.141592565);
        } else {
            scattering_pdf = (float)(1 / (4 * 3.1415926));
        }
        ray.o = rec.p; ray.d = dir; ray.tm = ray_time0; /* ray(rec.p, dir, r.time()) */
        o.e.kx = scattering_pdf * attenuation.x; o.e.ky = scattering_pdf * attenuation.y; o.e.kz = scattering_pdf * attenuation.z;
        o.e.rp = 1 / pdf;
    } else { /* diffuse_light (materials.cuh:151-163) or unknown tag: no scatter */
        SHMARK("emissive");
        V3 emission = mk(0, 0, 0);
        if (mtype == MORT_MAT_DIFFUSE_LIGHT && rec.front_face)
            emission = lambert_color_rec(sc, sc.dlight[midx], rec);
        o.final_value = emission;
        o.done = true;
    }
    SHMARK("shade_end");
    return o;
}

#endif
