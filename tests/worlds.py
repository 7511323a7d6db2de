"""Hand-built brute-force worlds (no reference BVH) shared by the GPU parity tests (tests/test_gpu_gen.py) and the
CPU tests of the host loop (tests/test_host_mode.py): the awkward cases the built-in scenes do not contain."""
import ctypes as C

from mort_amd import host, structs as S


def set_view(cam, frm, at, vfov=None, defocus=None):
    for i in range(3):
        cam.lookfrom.e[i] = frm[i]; cam.lookat.e[i] = at[i]
    if vfov is not None:
        cam.vfov = vfov
    if defocus is not None:
        cam.defocus_angle = defocus
    host.lib().mort_camera_initialize(C.byref(cam))


def flat_world(prims, media=(), light=None, late_list=False):
    """A brute-force world (no BVH).  prims: ("sphere", centre, radius, mat) | ("msphere", c1, c2, radius, mat) |
    ("quad", Q, u, v, mat) | ("box", a, b, mat) | ("rbox", size, translation, theta, mat); mat as in
    test_gpu_parity._custom_bvh_world.  media: (centre, radius, density, rgb).  Returns (world, ids of the prims)."""
    L = host.lib()
    w = host.World()

    def material(mat):
        kind = mat[0]
        if kind in ("lamb", "light", "iso"):
            col = L.mort_add_solid_color(w.ptr, host.vec3(*mat[1]))
            add = {"lamb": L.mort_add_lambertian, "light": L.mort_add_diffuse_light, "iso": L.mort_add_isotropic}[kind]
            return {"lamb": S.MAT_LAMBERTIAN, "light": S.MAT_DIFFUSE_LIGHT, "iso": S.MAT_ISOTROPIC}[kind], add(w.ptr, S.TEXTURE_SOLID, col)
        if kind == "checker":
            c1 = L.mort_add_solid_color(w.ptr, host.vec3(.2, .3, .1)); c2 = L.mort_add_solid_color(w.ptr, host.vec3(.9, .9, .9))
            return S.MAT_LAMBERTIAN, L.mort_add_lambertian(w.ptr, S.TEXTURE_CHECKER, L.mort_add_checker_texture(w.ptr, 0.32, S.TEXTURE_SOLID, c1, S.TEXTURE_SOLID, c2))
        if kind == "noise":
            g = S.HostRng(); L.mort_host_rng_init(C.byref(g), 3, 0)
            return S.MAT_LAMBERTIAN, L.mort_add_lambertian(w.ptr, S.TEXTURE_NOISE, L.mort_add_noise_texture(w.ptr, 4.0, C.byref(g)))
        if kind == "image":
            img = host.synthetic_earth(64, 32); w._keepalive.append(img)
            return S.MAT_LAMBERTIAN, L.mort_add_lambertian(w.ptr, S.TEXTURE_IMAGE, L.mort_add_image_texture(w.ptr, img.ctypes.data, 64, 32))
        if kind == "metal":
            return S.MAT_METAL, L.mort_add_metal(w.ptr, host.vec3(*mat[1]), mat[2])
        return S.MAT_DIELECTRIC, L.mort_add_dielectric(w.ptr, mat[1])

    ids = []
    for p in prims:
        mt, mi = material(p[-1])
        if p[0] == "sphere":
            ids.append((S.OBJ_SPHERE, L.mort_add_sphere(w.ptr, host.vec3(*p[1]), p[2], mt, mi, False)))
        elif p[0] == "msphere":
            ids.append((S.OBJ_SPHERE, L.mort_add_moving_sphere(w.ptr, host.vec3(*p[1]), host.vec3(*p[2]), p[3], mt, mi, False)))
        elif p[0] == "quad":
            ids.append((S.OBJ_QUAD, L.mort_add_quad(w.ptr, host.vec3(*p[1]), host.vec3(*p[2]), host.vec3(*p[3]), mt, mi, False)))
        elif p[0] == "box":
            L.mort_box(w.ptr, host.vec3(*p[1]), host.vec3(*p[2]), mt, mi); ids.append(None)
        else:
            L.mort_rotated_box(w.ptr, host.vec3(*p[1]), host.vec3(*p[2]), p[3], mt, mi); ids.append(None)
    for centre, radius, density, rgb in media:
        b = L.mort_add_sphere(w.ptr, host.vec3(*centre), radius, S.MAT_DIELECTRIC, L.mort_add_dielectric(w.ptr, 1.5), True)
        col = L.mort_add_solid_color(w.ptr, host.vec3(*rgb))
        L.mort_add_constant_medium(w.ptr, S.OBJ_SPHERE, b, density, S.MAT_ISOTROPIC, L.mort_add_isotropic(w.ptr, S.TEXTURE_SOLID, col), False)
    if late_list:  # a non-skip list is scanned AFTER the media (world.cuh:154-168): order matters, no unified tree
        lst = L.mort_add_hittable_list(w.ptr, False)
        mt, mi = material(("lamb", (.4, .4, .9)))
        L.mort_list_add(w.ptr, lst, S.OBJ_SPHERE, L.mort_add_sphere(w.ptr, host.vec3(0.2, 0.1, -1.2), 0.35, mt, mi, True))
    w.c.bvh_mode = False
    return w, ids


def flat_camera(light=None, spp=9, width=144, depth=12):
    _, cam = host.build_scene(2, width=width, spp=spp, depth=depth)
    cam.background.e[0], cam.background.e[1], cam.background.e[2] = 0.30, 0.35, 0.45
    if light is not None:
        cam.light_obj_type, cam.light_obj_idx = light
    set_view(cam, (0.0, 0.8, 2.5), (0.0, 0.0, -1.0), vfov=55, defocus=0.0)
    return cam


GROUND = ("sphere", (0, -100.5, -1), 100, ("checker",))
FLAT_WORLDS = {
    "empty": dict(prims=[]),
    "single_sphere": dict(prims=[("sphere", (0, 0, -1), 0.5, ("lamb", (.7, .3, .3)))]),
    "single_quad": dict(prims=[("quad", (-1, -0.5, -1.5), (2, 0, 0), (0, 1.5, 0.3), ("metal", (.8, .8, .8), 0.05))]),
    # equal t everywhere: the same sphere / quad several times, a quad lying in another one's plane -> the scan decides
    "coincident": dict(prims=[GROUND] + [("sphere", (0, 0, -1), 0.5, ("lamb", (.1 * k, .2, .5))) for k in range(4)] +
                       [("quad", (-2, -0.5, -2), (4, 0, 0), (0, 2, 0), ("lamb", (.2 * k, .5, .2))) for k in range(3)] +
                       [("quad", (-1, 0, -2), (1, 0, 0), (0, 1, 0), ("metal", (.9, .9, .2), 0.0))]),
    "concentric_glass": dict(prims=[GROUND, ("sphere", (0, 0, -1), 0.5, ("glass", 1.5)), ("sphere", (0, 0, -1), 0.4, ("glass", 1 / 1.5)),
                                    ("sphere", (0, 0, -1), 0.2, ("glass", 1.5)), ("msphere", (-1.2, 0, -1), (-1.2, 0.4, -1), 0.4, ("lamb", (.8, .8, 0))),
                                    ("sphere", (1.2, 0, -1), 0.5, ("metal", (.8, .6, .2), 1.0))]),
    "boxes_and_instances": dict(prims=[GROUND, ("box", (-1.5, -0.5, -2.0), (-0.7, 0.4, -1.2), ("lamb", (.7, .2, .2))),
                                       ("rbox", (0.8, 1.2, 0.8), (0.2, -0.5, -2.2), 25.0, ("metal", (.7, .7, .7), 0.1)),
                                       ("rbox", (0.5, 0.5, 0.5), (-0.3, -0.5, -0.8), -40.0, ("glass", 1.5))]),
    # isotropic and emissive spheres UNDER A LIGHT OBJECT (mixture pdf + sphere_pdf, pdf.cuh:29-37,91-103)
    "every_material_lit_by_sphere": dict(prims=[GROUND] + [("sphere", (-3 + 1.0 * k, 0.0, -1.5 - 0.2 * k), 0.45, m) for k, m in enumerate(
        [("lamb", (.7, .3, .3)), ("noise",), ("image",), ("metal", (.7, .6, .5), 0.2), ("glass", 1.5), ("light", (6, 6, 6)), ("iso", (.3, .6, .9))])],
        light=("prim", 6)),  # the emissive sphere (world sphere index 6) is the light object
    "lit_by_quad_with_media": dict(prims=[GROUND, ("quad", (-1, 2.2, -2), (2, 0, 0), (0, 0, 2), ("light", (9, 9, 9))),
                                          ("sphere", (-0.9, 0, -1), 0.5, ("iso", (.9, .4, .2))), ("sphere", (0.9, 0, -1), 0.5, ("lamb", (.2, .4, .9)))],
                                   media=[((0, 0.2, -1), 0.6, 1.5, (.9, .9, .9)), ((0, 0, 0), 30.0, 0.02, (1, 1, 1))], light=("prim", 1)),
    "media_then_list": dict(prims=[GROUND, ("sphere", (-0.9, 0, -1), 0.5, ("lamb", (.9, .4, .2)))],
                            media=[((0, 0.2, -1), 0.7, 2.0, (.9, .9, .9))], late_list=True),
}




def custom_bvh_world(spheres, noise_seed=1):
    """A world that is one BVH over the given spheres: (centre, radius, material) with material one of
    ("lamb", rgb) ("checker",) ("noise",) ("image",) ("metal", rgb, fuzz) ("glass", ior) ("light", rgb) ("iso", rgb)."""
    import ctypes as C
    L = host.lib()
    w = host.World()
    lst = L.mort_add_hittable_list(w.ptr, True)
    LAMB, METAL, DIEL, LIGHT, ISO = S.MAT_LAMBERTIAN, S.MAT_METAL, S.MAT_DIELECTRIC, S.MAT_DIFFUSE_LIGHT, S.MAT_ISOTROPIC
    for centre, radius, mat in spheres:
        kind = mat[0]
        if kind in ("lamb", "light", "iso"):
            col = L.mort_add_solid_color(w.ptr, host.vec3(*mat[1]))
            add = {"lamb": L.mort_add_lambertian, "light": L.mort_add_diffuse_light, "iso": L.mort_add_isotropic}[kind]
            mt, mi = {"lamb": LAMB, "light": LIGHT, "iso": ISO}[kind], add(w.ptr, S.TEXTURE_SOLID, col)
        elif kind == "checker":
            c1 = L.mort_add_solid_color(w.ptr, host.vec3(.2, .3, .1)); c2 = L.mort_add_solid_color(w.ptr, host.vec3(.9, .9, .9))
            ck = L.mort_add_checker_texture(w.ptr, 0.32, S.TEXTURE_SOLID, c1, S.TEXTURE_SOLID, c2)
            mt, mi = LAMB, L.mort_add_lambertian(w.ptr, S.TEXTURE_CHECKER, ck)
        elif kind == "noise":
            g = S.HostRng(); L.mort_host_rng_init(C.byref(g), noise_seed, 0)
            nz = L.mort_add_noise_texture(w.ptr, 4.0, C.byref(g))
            mt, mi = LAMB, L.mort_add_lambertian(w.ptr, S.TEXTURE_NOISE, nz)
        elif kind == "image":
            img = host.synthetic_earth(64, 32); w._keepalive.append(img)
            im = L.mort_add_image_texture(w.ptr, img.ctypes.data, 64, 32)
            mt, mi = LAMB, L.mort_add_lambertian(w.ptr, S.TEXTURE_IMAGE, im)
        elif kind == "metal":
            mt, mi = METAL, L.mort_add_metal(w.ptr, host.vec3(*mat[1]), mat[2])
        else:
            mt, mi = DIEL, L.mort_add_dielectric(w.ptr, mat[1])
        L.mort_list_add(w.ptr, lst, S.OBJ_SPHERE, L.mort_add_sphere(w.ptr, host.vec3(*centre), radius, mt, mi, True))
    L.mort_add_bvh(w.ptr, lst, False)
    w.c.bvh_mode = True
    return w



_BVH_GROUND = ((0, -100.5, -1), 100, ("checker",))
BVH_WORLDS = {
    "one": [((0, 0, -1), 0.5, ("lamb", (.7, .3, .3)))],
    "two": [_BVH_GROUND, ((0, 0, -1), 0.5, ("glass", 1.5))],
    "three": [_BVH_GROUND, ((0, 0, -1), 0.5, ("lamb", (.1, .2, .5))), ((1, 0, -1), 0.5, ("metal", (.8, .6, .2), 0.3))],
    "coincident": [_BVH_GROUND] + [((0, 0, -1), 0.5, ("lamb", (.1 * k, .2, .5))) for k in range(5)] + [((0.3, 0, -1), 0.5, ("metal", (.8, .8, .8), 0.0))] * 3,
    "concentric_glass": [_BVH_GROUND, ((0, 0, -1), 0.5, ("glass", 1.5)), ((0, 0, -1), 0.4, ("glass", 1.0 / 1.5)), ((0, 0, -1), 0.2, ("glass", 1.5)),
                         ((-1.1, 0, -1), 0.5, ("lamb", (.8, .8, 0))), ((1.1, 0, -1), 0.5, ("metal", (.8, .6, .2), 1.0))],
    "every_material": [_BVH_GROUND] + [((-3 + 1.0 * k, 0.0, -1.5 - 0.2 * k), 0.45, m) for k, m in enumerate(
        [("lamb", (.7, .3, .3)), ("noise",), ("image",), ("metal", (.7, .6, .5), 0.2), ("glass", 1.5), ("light", (4, 4, 4)), ("iso", (.3, .6, .9))])],
}
