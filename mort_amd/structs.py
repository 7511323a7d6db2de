"""ctypes mirrors of include/mort_scene.h (which mirrors the reference structs:
objects.cuh:147-161,237-249,280-287,368-375,440-448,510-518,725-735;
materials.cuh:28-202; textures.cuh:17-266; camera.cuh:13-45; world.cuh:173-178).

Sizes are asserted against the C header's static asserts.
"""
import ctypes as C

LIST_MAX_OBJS = 1000
MAX_BVH_NODES = 1024
POINT_COUNT = 256
MAX_BOUNCE_LIMIT = 64

OBJ_SPHERE, OBJ_QUAD, OBJ_TRANSLATE, OBJ_ROTATE_Y, OBJ_CONSTANT_MEDIUM, OBJ_HITTABLE_LIST, OBJ_BVH = range(1, 8)
MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_DIFFUSE_LIGHT, MAT_ISOTROPIC = range(1, 6)
TEXTURE_SOLID, TEXTURE_CHECKER, TEXTURE_IMAGE, TEXTURE_NOISE = range(1, 5)

DEFAULT_SEED = 69420  # mort.cu:707


class Vec3(C.Structure):
    _fields_ = [("e", C.c_float * 3)]

    def tolist(self):
        return [self.e[0], self.e[1], self.e[2]]


class Interval(C.Structure):
    _fields_ = [("imin", C.c_float), ("imax", C.c_float)]


class Aabb(C.Structure):
    _fields_ = [("x", Interval), ("y", Interval), ("z", Interval)]


class Sphere(C.Structure):
    _fields_ = [("center1", Vec3), ("radius", C.c_float), ("moves", C.c_bool), ("center_vec", Vec3),
                ("mat_type", C.c_int), ("mat_idx", C.c_int), ("idx", C.c_int), ("skip", C.c_bool), ("bbox", Aabb)]


class Quad(C.Structure):
    _fields_ = [("Q", Vec3), ("u", Vec3), ("v", Vec3), ("normal", Vec3), ("w", Vec3), ("bbox", Aabb),
                ("D", C.c_float), ("area", C.c_float), ("mat_type", C.c_int), ("mat_idx", C.c_int),
                ("idx", C.c_int), ("skip", C.c_bool)]


class Translate(C.Structure):
    _fields_ = [("obj_type", C.c_int), ("obj_idx", C.c_int), ("offset", Vec3), ("bbox", Aabb),
                ("idx", C.c_int), ("skip", C.c_bool)]


class RotateY(C.Structure):
    _fields_ = [("obj_type", C.c_int), ("obj_idx", C.c_int), ("sin_theta", C.c_float), ("cos_theta", C.c_float),
                ("bbox", Aabb), ("idx", C.c_int), ("skip", C.c_bool)]


class ConstantMedium(C.Structure):
    _fields_ = [("obj_type", C.c_int), ("obj_idx", C.c_int), ("neg_inv_density", C.c_double),
                ("mat_type", C.c_int), ("mat_idx", C.c_int), ("bbox", Aabb), ("idx", C.c_int), ("skip", C.c_bool)]


class HittableList(C.Structure):
    _fields_ = [("obj_types", C.c_int * LIST_MAX_OBJS), ("obj_idxs", C.c_int * LIST_MAX_OBJS),
                ("num_objs", C.c_int), ("idx", C.c_int), ("skip", C.c_bool), ("bbox", Aabb)]


class Bvh(C.Structure):
    _fields_ = [("left_children_types", C.c_int * MAX_BVH_NODES), ("left_children_idxs", C.c_int * MAX_BVH_NODES),
                ("right_children_types", C.c_int * MAX_BVH_NODES), ("right_children_idxs", C.c_int * MAX_BVH_NODES),
                ("is_internal_node", C.c_bool * MAX_BVH_NODES), ("bounding_boxes", Aabb * MAX_BVH_NODES),
                ("idx", C.c_int), ("skip", C.c_bool)]


class Lambertian(C.Structure):
    _fields_ = [("texType", C.c_int), ("texIdx", C.c_int), ("idx", C.c_int)]


class Metal(C.Structure):
    _fields_ = [("albedo", Vec3), ("fuzz", C.c_float), ("idx", C.c_int)]


class Dielectric(C.Structure):
    _fields_ = [("ior", C.c_float), ("inv_ior", C.c_float), ("albedo", Vec3), ("idx", C.c_int)]


class DiffuseLight(C.Structure):
    _fields_ = [("texType", C.c_int), ("texIdx", C.c_int), ("idx", C.c_int)]


class Isotropic(C.Structure):
    _fields_ = [("texType", C.c_int), ("texIdx", C.c_int), ("idx", C.c_int)]


class SolidColor(C.Structure):
    _fields_ = [("color_value", Vec3), ("idx", C.c_int)]


class CheckerTexture(C.Structure):
    _fields_ = [("inv_scale", C.c_float), ("evenTextureType", C.c_int), ("evenTextureIdx", C.c_int),
                ("oddTextureType", C.c_int), ("oddTextureIdx", C.c_int), ("idx", C.c_int)]


class ImageTexture(C.Structure):
    _fields_ = [("texels", C.c_void_p), ("width", C.c_int), ("height", C.c_int), ("idx", C.c_int)]


class NoiseTexture(C.Structure):
    _fields_ = [("ranvec", Vec3 * POINT_COUNT), ("perm_x", C.c_int * POINT_COUNT), ("perm_y", C.c_int * POINT_COUNT),
                ("perm_z", C.c_int * POINT_COUNT), ("scale", C.c_float), ("idx", C.c_int)]


class WorldObjects(C.Structure):
    _fields_ = [("host_sphere", C.POINTER(Sphere)), ("num_spheres", C.c_int),
                ("host_quad", C.POINTER(Quad)), ("num_quads", C.c_int),
                ("host_translate", C.POINTER(Translate)), ("num_translates", C.c_int),
                ("host_rotate_y", C.POINTER(RotateY)), ("num_rotate_y", C.c_int),
                ("host_constant_medium", C.POINTER(ConstantMedium)), ("num_constant_medium", C.c_int),
                ("host_hittable_list", C.POINTER(HittableList)), ("num_hittable_list", C.c_int),
                ("host_bvh", C.POINTER(Bvh)), ("num_bvh", C.c_int)]


class WorldMaterials(C.Structure):
    _fields_ = [("host_lambertian", C.POINTER(Lambertian)), ("num_lambertians", C.c_int),
                ("host_metal", C.POINTER(Metal)), ("num_metals", C.c_int),
                ("host_dielectric", C.POINTER(Dielectric)), ("num_dielectrics", C.c_int),
                ("host_diffuse_light", C.POINTER(DiffuseLight)), ("num_diffuse_lights", C.c_int),
                ("host_isotropic", C.POINTER(Isotropic)), ("num_isotropics", C.c_int)]


class WorldTextures(C.Structure):
    _fields_ = [("host_solid_color", C.POINTER(SolidColor)), ("num_solid_colors", C.c_int),
                ("host_checker_texture", C.POINTER(CheckerTexture)), ("num_checker_textures", C.c_int),
                ("host_image_texture", C.POINTER(ImageTexture)), ("num_image_textures", C.c_int),
                ("host_noise_texture", C.POINTER(NoiseTexture)), ("num_noise_textures", C.c_int)]


class World(C.Structure):
    _fields_ = [("objs", WorldObjects), ("mats", WorldMaterials), ("texs", WorldTextures), ("bvh_mode", C.c_bool)]


class Camera(C.Structure):
    _fields_ = [("aspect_ratio", C.c_float), ("image_width", C.c_int), ("image_height", C.c_int),
                ("samples_per_pixel", C.c_int), ("pixel_samples_scale", C.c_float), ("sqrt_spp", C.c_int),
                ("recip_sqrt_spp", C.c_float), ("bounce_limit", C.c_int), ("vfov", C.c_int), ("background", Vec3),
                ("recursionAttenuation", C.c_void_p), ("recursionEmission", C.c_void_p),
                ("recursionScatteringPdf", C.c_void_p), ("recursionPdf", C.c_void_p),
                ("light_obj_type", C.c_int), ("light_obj_idx", C.c_int),
                ("center", Vec3), ("pixel00_loc", Vec3), ("pixel_delta_u", Vec3), ("pixel_delta_v", Vec3),
                ("lookfrom", Vec3), ("lookat", Vec3), ("vup", Vec3), ("v", Vec3), ("u", Vec3), ("w", Vec3),
                ("defocus_angle", C.c_float), ("focus_dist", C.c_float),
                ("defocus_disk_u", Vec3), ("defocus_disk_v", Vec3)]


class RngState(C.Structure):
    _fields_ = [("d", C.c_uint), ("v", C.c_uint * 5), ("boxmuller_flag", C.c_int),
                ("boxmuller_flag_double", C.c_int), ("boxmuller_extra", C.c_float),
                ("boxmuller_extra_double", C.c_double)]


class HostRng(C.Structure):
    _fields_ = [("state", C.c_uint32), ("args_rtl", C.c_int)]


class SceneOpts(C.Structure):
    _fields_ = [("args_rtl", C.c_int), ("earth_texels", C.c_void_p), ("earth_width", C.c_int),
                ("earth_height", C.c_int)]


_SIZES = {Sphere: 72, Quad: 108, Translate: 52, RotateY: 48, ConstantMedium: 56, HittableList: 8036, Bvh: 41992,
          Lambertian: 12, Metal: 20, Dielectric: 24, DiffuseLight: 12, Isotropic: 12, SolidColor: 16,
          CheckerTexture: 24, ImageTexture: 24, NoiseTexture: 6152, World: 264, Camera: 240, RngState: 48}
for _t, _n in _SIZES.items():
    assert C.sizeof(_t) == _n, (_t.__name__, C.sizeof(_t), _n)
