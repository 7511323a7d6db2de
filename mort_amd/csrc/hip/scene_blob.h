/*
 * scene_blob.h -- world -> one blob of flat arrays + the DScene view over it.  Shared by mort_hip_upload_world (blob in
 * HBM) and the host loop of mort_hip_render_host (blob in host memory): validation of every index the kernels will
 * dereference, the scene compiler (scene_compile.h), and the placement of its arrays.  Replaces world::toDevice()
 * (world.cuh:98-102).
 */
#ifndef MORT_SCENE_BLOB_H
#define MORT_SCENE_BLOB_H

#include <cstring>
#include <vector>

#include "scene_compile.h"
#include "dev_render.h"

/* ---- validation of every index the kernels will dereference ---- */
static inline bool tex_ok(const mort_world *w, int type, int idx, int depth) {
    switch (type) {
    case MORT_TEXTURE_SOLID: return idx >= 0 && idx < w->texs.num_solid_colors;
    case MORT_TEXTURE_CHECKER: {
        if (idx < 0 || idx >= w->texs.num_checker_textures || depth > 6) return false;
        const mort_checker_texture &c = w->texs.host_checker_texture[idx];
        return tex_ok(w, c.evenTextureType, c.evenTextureIdx, depth + 1) && tex_ok(w, c.oddTextureType, c.oddTextureIdx, depth + 1);
    }
    case MORT_TEXTURE_IMAGE: return idx >= 0 && idx < w->texs.num_image_textures;
    case MORT_TEXTURE_NOISE: return idx >= 0 && idx < w->texs.num_noise_textures;
    }
    return true; /* unknown tag: the error pattern, no table access */
}
static inline bool mat_ok(const mort_world *w, int type, int idx) {
    const mort_world_materials &m = w->mats;
    switch (type) {
    case MORT_MAT_LAMBERTIAN: return idx >= 0 && idx < m.num_lambertians && tex_ok(w, m.host_lambertian[idx].texType, m.host_lambertian[idx].texIdx, 0);
    case MORT_MAT_METAL: return idx >= 0 && idx < m.num_metals;
    case MORT_MAT_DIELECTRIC: return idx >= 0 && idx < m.num_dielectrics;
    case MORT_MAT_DIFFUSE_LIGHT: return idx >= 0 && idx < m.num_diffuse_lights && tex_ok(w, m.host_diffuse_light[idx].texType, m.host_diffuse_light[idx].texIdx, 0);
    case MORT_MAT_ISOTROPIC: return idx >= 0 && idx < m.num_isotropics && tex_ok(w, m.host_isotropic[idx].texType, m.host_isotropic[idx].texIdx, 0);
    }
    return true; /* unknown tag: treated as "does not scatter", no table access */
}
static inline int validate_world(const mort_world *w) {
    const mort_world_objects &o = w->objs;
    if (o.num_spheres < 0 || o.num_spheres > MORT_NUM_SPHERES || o.num_quads < 0 || o.num_quads > MORT_NUM_QUADS ||
        o.num_translates < 0 || o.num_translates > MORT_NUM_TRANSLATE || o.num_rotate_y < 0 || o.num_rotate_y > MORT_NUM_ROTATE_Y ||
        o.num_constant_medium < 0 || o.num_constant_medium > MORT_NUM_CONSTANT_MEDIUM ||
        o.num_hittable_list < 0 || o.num_hittable_list > MORT_NUM_HITTABLE_LIST || o.num_bvh < 0 || o.num_bvh > MORT_NUM_BVH)
        return MORT_ERR_CAPACITY;
    for (int i = 0; i < o.num_spheres; i++) if (!mat_ok(w, o.host_sphere[i].mat_type, o.host_sphere[i].mat_idx)) return MORT_ERR_INVALID;
    for (int i = 0; i < o.num_quads; i++) if (!mat_ok(w, o.host_quad[i].mat_type, o.host_quad[i].mat_idx)) return MORT_ERR_INVALID;
    for (int i = 0; i < o.num_constant_medium; i++) if (!mat_ok(w, o.host_constant_medium[i].mat_type, o.host_constant_medium[i].mat_idx)) return MORT_ERR_INVALID;
    for (int i = 0; i < o.num_hittable_list; i++) if (o.host_hittable_list[i].num_objs < 0 || o.host_hittable_list[i].num_objs > MORT_LIST_MAX_OBJS) return MORT_ERR_INVALID;
    return MORT_OK;
}

template <typename T>
static inline size_t place(std::vector<unsigned char> &blob, const std::vector<T> &v) {
    size_t off = (blob.size() + 15) & ~(size_t)15;
    blob.resize(off + v.size() * sizeof(T) + 16, 0); /* 16 B tail so empty arrays still get distinct, valid addresses */
    if (!v.empty()) std::memcpy(blob.data() + off, v.data(), v.size() * sizeof(T));
    return off;
}


struct SceneBlob {
    mortc::Compiled comp;
    std::vector<unsigned char> bytes;
    size_t o_items, o_sub, o_nodes, o_sph, o_quads, o_xf, o_media, o_lamb, o_metal, o_diel, o_dl, o_iso, o_solid, o_chk, o_img;
    size_t hot_bytes, o_wsph, o_wquads, o_lt, o_li, o_lf, o_lc, o_noise, o_tex;
};

/* MORT_OK, or the status upload_world returns for this world */
static inline int build_scene_blob(const mort_world *w, SceneBlob &sb) {
    int st = validate_world(w);
    if (st != MORT_OK) return st;
    mortc::Compiler comp;
    comp.w = w;
    comp.run();
    if (comp.out.status != MORT_OK) return comp.out.status;
    if (comp.out.inverted_box) return MORT_ERR_UNSUPPORTED; /* slab_hit orders planes with min/max: needs min <= max boxes */
    sb.comp = std::move(comp.out);
    const mortc::Compiled &o = sb.comp;
    std::vector<unsigned char> &blob = sb.bytes;
    blob.clear();
    sb.o_items = place(blob, o.items); sb.o_sub = place(blob, o.subitems); sb.o_nodes = place(blob, o.nodes);
    sb.o_sph = place(blob, o.spheres); sb.o_quads = place(blob, o.quads); sb.o_xf = place(blob, o.xforms);
    sb.o_media = place(blob, o.media);
    sb.o_lamb = place(blob, o.lambert); sb.o_metal = place(blob, o.metal); sb.o_diel = place(blob, o.dielectric);
    sb.o_dl = place(blob, o.dlight); sb.o_iso = place(blob, o.isotropic);
    sb.o_solid = place(blob, o.solid); sb.o_chk = place(blob, o.checker); sb.o_img = place(blob, o.image);
    sb.hot_bytes = (blob.size() + 15) & ~(size_t)15;
    sb.o_wsph = place(blob, o.wspheres); sb.o_wquads = place(blob, o.wquads);
    sb.o_lt = place(blob, o.list_types); sb.o_li = place(blob, o.list_idxs);
    sb.o_lf = place(blob, std::vector<int>(o.list_first, o.list_first + MORT_NUM_HITTABLE_LIST));
    sb.o_lc = place(blob, std::vector<int>(o.list_count, o.list_count + MORT_NUM_HITTABLE_LIST));
    sb.o_noise = place(blob, o.noise); sb.o_tex = place(blob, o.texels);
    return MORT_OK;
}

/* the DScene whose pointers address a copy of the blob at `base` (device or host memory) */
static inline void scene_view(const SceneBlob &sb, const unsigned char *base, DScene &s) {
    const mortc::Compiled &o = sb.comp;
    std::memset(&s, 0, sizeof s);
    s.items = (const DItem *)(base + sb.o_items); s.n_items = (int)o.items.size();
    s.subitems = (const DItem *)(base + sb.o_sub); s.n_subitems = (int)o.subitems.size();
    s.nodes = (const DBvhNode *)(base + sb.o_nodes); s.n_nodes = (int)o.nodes.size();
    s.spheres = (const DSphere *)(base + sb.o_sph); s.n_spheres = (int)o.spheres.size();
    s.quads = (const DQuad *)(base + sb.o_quads); s.n_quads = (int)o.quads.size();
    s.xforms = (const DXform *)(base + sb.o_xf); s.n_xforms = (int)o.xforms.size();
    s.neg_inv_density = (const double *)(base + sb.o_media); s.n_media = (int)o.media.size();
    s.lambert = (const DLambert *)(base + sb.o_lamb); s.metal = (const DMetal *)(base + sb.o_metal);
    s.dielectric = (const DDielectric *)(base + sb.o_diel);
    s.dlight = (const DLambert *)(base + sb.o_dl); s.isotropic = (const DLambert *)(base + sb.o_iso);
    s.solid = (const DSolid *)(base + sb.o_solid); s.checker = (const DChecker *)(base + sb.o_chk);
    s.image = (const DImage *)(base + sb.o_img); s.image_hbm = s.image;
    s.texels = base + sb.o_tex; s.noise = (const float *)(base + sb.o_noise);
    s.wspheres = (const DSphere *)(base + sb.o_wsph); s.wquads = (const DQuad *)(base + sb.o_wquads);
    s.list_types = (const int *)(base + sb.o_lt); s.list_idxs = (const int *)(base + sb.o_li);
    s.list_first = (const int *)(base + sb.o_lf); s.list_count = (const int *)(base + sb.o_lc);
    s.blob_bytes = (uint32_t)sb.hot_bytes;
    s.lds_bytes = (uint32_t)sb.hot_bytes;
}

/* the camera half of the launch arguments (camera.cuh:13-45 after initialize()) */
static inline V3 to_v3(const mort_vec3 &v) { V3 r; r.x = v.e[0]; r.y = v.e[1]; r.z = v.e[2]; return r; }
static inline void render_args_camera(RenderArgs &a, const mort_camera *cam) {
    a.width = cam->image_width; a.height = cam->image_height;
    a.sqrt_spp = cam->sqrt_spp; a.bounce_limit = cam->bounce_limit;
    a.recip_sqrt_spp = cam->recip_sqrt_spp; a.pixel_samples_scale = cam->pixel_samples_scale;
    a.background = to_v3(cam->background); a.center = to_v3(cam->center); a.pixel00 = to_v3(cam->pixel00_loc);
    a.du = to_v3(cam->pixel_delta_u); a.dv = to_v3(cam->pixel_delta_v);
    a.defocus_u = to_v3(cam->defocus_disk_u); a.defocus_v = to_v3(cam->defocus_disk_v);
    a.defocus_angle = cam->defocus_angle;
    a.light_type = cam->light_obj_type; a.light_idx = cam->light_obj_idx;
}

/* the camera's light object must name primitives the light-sampling code can index (pdf.cuh:60-80) */
static inline int check_light_object(const mortc::Compiled &o, int n_lists, int type, int idx) {
    const int n_wspheres = (int)o.wspheres.size(), n_wquads = (int)o.wquads.size();
    if (type == -1) return MORT_OK;
    if (type == MORT_OBJ_SPHERE) return (idx >= 0 && idx < n_wspheres) ? MORT_OK : MORT_ERR_INVALID;
    if (type == MORT_OBJ_QUAD) return (idx >= 0 && idx < n_wquads) ? MORT_OK : MORT_ERR_INVALID;
    if (type == MORT_OBJ_HITTABLE_LIST) {
        if (idx < 0 || idx >= n_lists || idx >= MORT_NUM_HITTABLE_LIST) return MORT_ERR_INVALID;
        if (o.list_count[idx] <= 0) return MORT_ERR_INVALID;
        for (int i = 0; i < o.list_count[idx]; i++) {
            const int t = o.list_types[o.list_first[idx] + i], k = o.list_idxs[o.list_first[idx] + i];
            if (t == MORT_OBJ_SPHERE) { if (k < 0 || k >= n_wspheres) return MORT_ERR_INVALID; }
            else if (t == MORT_OBJ_QUAD) { if (k < 0 || k >= n_wquads) return MORT_ERR_INVALID; }
            else if (t == MORT_OBJ_HITTABLE_LIST) return MORT_ERR_UNSUPPORTED; /* nested light lists */
        }
        return MORT_OK;
    }
    return MORT_OK; /* any other tag samples nothing: pdf 0, direction (1,0,0) (objects.cuh:961,978) */
}

#endif
