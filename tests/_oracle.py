"""ctypes binding of the CPU oracle (oracle/libdray_oracle.so).

Test infrastructure: the oracle is the checker for the HIP path and is loaded
only from tests/, the smoke check and the bench's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libdray_oracle.so")

D3 = C.c_double * 3


class Camera(C.Structure):
    _fields_ = [
        ("aspect_ratio", C.c_double),
        ("image_width", C.c_int32),
        ("samples_per_pixel", C.c_int32),
        ("max_depth", C.c_int32),
        ("_pad", C.c_int32),
        ("vfov", C.c_double),
        ("lookfrom", D3),
        ("lookat", D3),
        ("vup", D3),
        ("defocus_angle", C.c_double),
        ("focus_dist", C.c_double),
    ]


class CameraDerived(C.Structure):
    _fields_ = [
        ("image_height", C.c_int32),
        ("_pad", C.c_int32),
        ("pixel_samples_scale", C.c_double),
        ("center", D3),
        ("pixel00_loc", D3),
        ("pixel_delta_u", D3),
        ("pixel_delta_v", D3),
        ("u", D3),
        ("v", D3),
        ("w", D3),
        ("defocus_disk_u", D3),
        ("defocus_disk_v", D3),
        ("viewport_height", C.c_double),
        ("viewport_width", C.c_double),
        ("defocus_radius", C.c_double),
    ]


class Sphere(C.Structure):
    _fields_ = [
        ("center", D3),
        ("radius", C.c_double),
        ("kind", C.c_uint32),
        ("shape", C.c_uint32),
        ("albedo", D3),
        ("fuzz_or_ior", C.c_double),
    ]


SPHERE_DTYPE = np.dtype(
    [("center", "<f8", 3), ("radius", "<f8"), ("kind", "<u4"), ("shape", "<u4"), ("albedo", "<f8", 3),
     ("fuzz_or_ior", "<f8")]
)
assert SPHERE_DTYPE.itemsize == C.sizeof(Sphere) == 72


class Hit(C.Structure):
    _fields_ = [
        ("p", D3),
        ("normal", D3),
        ("t", C.c_double),
        ("front_face", C.c_int32),
        ("index", C.c_int32),
    ]


class Rng(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("key", C.c_uint32 * 2),
        ("pixel", C.c_uint32),
        ("sample", C.c_uint32),
        ("stage", C.c_uint32),
        ("script", C.POINTER(C.c_double)),
        ("script_len", C.c_size_t),
        ("script_pos", C.c_size_t),
        ("draws", C.c_uint64),
    ]


class Stats(C.Structure):
    _fields_ = [("segments", C.c_uint64), ("samples", C.c_uint64)]


PATH_ENTRY_DTYPE = np.dtype(
    [("attenuation", "<f8", 3), ("origin", "<f8", 3), ("direction", "<f8", 3), ("color", "<f8", 3), ("depth", "<i4"),
     ("stage", "<u4"), ("pixel_i", "<i4"), ("pixel_j", "<i4"), ("pixel_sample_num", "<i4"), ("finished", "<u4"),
     ("hit_object_or_stop", "<u4"), ("_pad", "<u4")])
assert PATH_ENTRY_DTYPE.itemsize == 128

MAT_DEFAULT, MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_TRANSPARENT = 0, 1, 2, 3, 4

_lib = None


def build():
    """(Re)build the oracle shared object with its Makefile."""
    subprocess.run(["make", "-C", ORACLE_DIR, "--no-print-directory"], check=True, stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(ORACLE_SO):
        build()
    L = C.CDLL(ORACLE_SO)
    P = C.POINTER
    dp = P(C.c_double)
    L.orc_philox4x32_10.argtypes = [P(C.c_uint32), P(C.c_uint32), P(C.c_uint32)]
    L.orc_u01.restype = C.c_double
    L.orc_u01.argtypes = [C.c_uint32]
    L.orc_rng_draw.restype = C.c_double
    L.orc_rng_draw.argtypes = [P(Rng), C.c_uint32]
    for name in ("add", "sub", "mul", "divv", "min", "max", "cross", "reflect"):
        getattr(L, "orc_v_" + name).argtypes = [dp, dp, dp]
    for name in ("scale", "div"):
        getattr(L, "orc_v_" + name).argtypes = [dp, C.c_double, dp]
    for name in ("neg", "unit"):
        getattr(L, "orc_v_" + name).argtypes = [dp, dp]
    L.orc_v_dot.restype = C.c_double
    L.orc_v_dot.argtypes = [dp, dp]
    for name in ("length", "length_squared"):
        getattr(L, "orc_v_" + name).restype = C.c_double
        getattr(L, "orc_v_" + name).argtypes = [dp]
    L.orc_v_near_zero.argtypes = [dp]
    L.orc_v_refract.argtypes = [dp, dp, C.c_double, dp]
    L.orc_random_unit_vector.argtypes = [P(Rng), dp]
    L.orc_random_in_unit_disk.argtypes = [P(Rng), dp]
    L.orc_surrounds.argtypes = [C.c_double] * 3
    L.orc_clamp.restype = C.c_double
    L.orc_clamp.argtypes = [C.c_double] * 3
    L.orc_sphere_hit.argtypes = [P(Sphere), dp, dp, C.c_double, C.c_double, P(Hit)]
    L.orc_list_hit.argtypes = [P(Sphere), C.c_size_t, dp, dp, C.c_double, C.c_double, P(Hit)]
    L.orc_box_hit.argtypes = [C.c_void_p, dp, dp, C.c_double, C.c_double, P(Hit)]
    L.orc_scatter.argtypes = [P(Sphere), dp, P(Hit), P(Rng), dp, dp, dp]
    L.orc_reflectance.restype = C.c_double
    L.orc_reflectance.argtypes = [C.c_double, C.c_double]
    L.orc_camera_default.argtypes = [P(Camera)]
    L.orc_camera_initialize.argtypes = [P(Camera), P(CameraDerived)]
    L.orc_get_ray.argtypes = [P(Camera), P(CameraDerived), C.c_int32, C.c_int32, P(Rng), dp, dp]
    for name in ("orc_ray_color_recursive", "orc_ray_color_iterative"):
        getattr(L, name).argtypes = [P(Sphere), C.c_size_t, dp, dp, C.c_int32, P(Rng), dp, P(C.c_uint64)]
    L.orc_linear_to_gamma.restype = C.c_double
    L.orc_linear_to_gamma.argtypes = [C.c_double]
    L.orc_color_to_rgb.argtypes = [dp, P(C.c_uint32)]
    L.orc_write_color.restype = C.c_uint32
    L.orc_write_color.argtypes = [C.c_int32, C.c_int32, C.c_size_t, dp, P(C.c_uint32), dp, P(C.c_int32)]
    L.orc_render.argtypes = [P(Camera), P(Sphere), C.c_size_t, P(C.c_uint32), dp, P(C.c_int32), C.c_size_t,
                             C.c_uint64, C.c_uint32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, P(Stats)]
    L.orc_scene_three_spheres.argtypes = [P(Sphere), C.c_size_t]
    L.orc_scene_final.argtypes = [C.c_uint64, P(Sphere), C.c_size_t]
    L.orc_scene_recipe.argtypes = [C.c_uint64, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, P(Sphere),
                                   C.c_size_t]
    L.orc_camera_ray_entries.argtypes = [P(Camera), C.c_uint64, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p]
    L.orc_ray_color_iteration_batch.argtypes = [C.c_void_p, C.c_size_t, P(Sphere), C.c_size_t, C.c_uint64, C.c_int32,
                                                C.c_uint32]
    _lib = L
    return L


# ---------------------------------------------------------------- helpers --

def v3(x, y, z):
    return D3(float(x), float(y), float(z))


def vec_op(name, a, b=None, s=None):
    L = lib()
    out = D3()
    fn = getattr(L, "orc_v_" + name)
    if b is not None:
        fn(v3(*a), v3(*b), out)
    elif s is not None:
        fn(v3(*a), float(s), out)
    else:
        fn(v3(*a), out)
    return tuple(out)


def script_rng(values):
    arr = (C.c_double * max(1, len(values)))(*values)
    r = Rng()
    r.kind = 1
    r.script = C.cast(arr, C.POINTER(C.c_double))
    r.script_len = len(values)
    r.script_pos = 0
    r._keepalive = arr
    return r


def philox_rng(seed, pixel, sample, stage):
    r = Rng()
    r.kind = 0
    r.key[0] = seed & 0xFFFFFFFF
    r.key[1] = (seed >> 32) & 0xFFFFFFFF
    r.pixel, r.sample, r.stage = pixel, sample, stage
    return r


def make_sphere(center, radius, kind, albedo=(0.0, 0.0, 0.0), fuzz_or_ior=0.0):
    s = Sphere()
    s.center = v3(*center)
    s.radius = max(float(radius), 0.0)
    s.kind = kind
    s.albedo = v3(*albedo)
    s.fuzz_or_ior = min(float(fuzz_or_ior), 1.0) if kind == MAT_METAL else float(fuzz_or_ior)
    return s


class Box(C.Structure):
    """orc_box: a BoundingBox record in an orc_sphere's place."""
    _fields_ = [("min", D3), ("_unused0", C.c_double), ("_unused_kind", C.c_uint32), ("shape", C.c_uint32),
                ("max", D3), ("_unused1", C.c_double)]


assert C.sizeof(Box) == C.sizeof(Sphere)
SHAPE_SPHERE, SHAPE_BOX = 0, 1


def make_box(x_min, x_max, y_min, y_max, z_min, z_max):
    """BoundingBox::new_xyz (bounding_box.rs:20-28) as a list record (returned as a Sphere so it can sit in a Sphere array)."""
    s = Sphere()
    s.center = v3(x_min, y_min, z_min)
    s.albedo = v3(x_max, y_max, z_max)
    s.shape = SHAPE_BOX
    return s


def spheres_to_numpy(arr, n):
    return np.frombuffer(arr, dtype=SPHERE_DTYPE, count=n).copy()


def numpy_to_spheres(a):
    a = np.ascontiguousarray(a, dtype=SPHERE_DTYPE)
    arr = (Sphere * len(a))()
    C.memmove(arr, a.ctypes.data, a.nbytes)
    return arr


def scene_three_spheres():
    arr = (Sphere * 4)()
    n = lib().orc_scene_three_spheres(arr, 4)
    assert n == 4
    return spheres_to_numpy(arr, n)


def scene_final(seed):
    arr = (Sphere * 600)()
    n = lib().orc_scene_final(seed, arr, 600)
    assert n > 0
    return spheres_to_numpy(arr, n)


def scene_recipe(seed, lo, hi, p_diffuse, p_metal, fuzz_max):
    cap = (hi - lo) * (hi - lo) + 8
    arr = (Sphere * cap)()
    n = lib().orc_scene_recipe(seed, lo, hi, p_diffuse, p_metal, fuzz_max, arr, cap)
    assert n > 0
    return spheres_to_numpy(arr, n)


def default_camera():
    cam = Camera()
    lib().orc_camera_default(C.byref(cam))
    return cam


def camera_three_spheres(width=400, spp=10, depth=10):
    cam = default_camera()
    cam.aspect_ratio = 16.0 / 9.0
    cam.image_width = width
    cam.samples_per_pixel = spp
    cam.max_depth = depth
    return cam


def camera_final(width=1200, spp=500, depth=50):
    """Camera of the reference's local render binary (src/main.rs:55-66)."""
    cam = default_camera()
    cam.aspect_ratio = 16.0 / 9.0
    cam.image_width = width
    cam.samples_per_pixel = spp
    cam.max_depth = depth
    cam.vfov = 20.0
    cam.lookfrom = v3(13.0, 2.0, 3.0)
    cam.lookat = v3(0.0, 0.0, 0.0)
    cam.vup = v3(0.0, 1.0, 0.0)
    cam.defocus_angle = 0.6
    cam.focus_dist = 10.0
    return cam


def camera_initialize(cam):
    d = CameraDerived()
    lib().orc_camera_initialize(C.byref(cam), C.byref(d))
    return d


def image_height(cam):
    return camera_initialize(cam).image_height


def render(cam, spheres, seed=1, sample_offset=0, iterative=0, row_begin=0, row_step=1, threads=1, color=None,
           raw=None, count=None):
    """Run the oracle render; returns (color u32[H,W], raw f64[H,W,3], count i32[H,W], stats)."""
    L = lib()
    H, W = image_height(cam), cam.image_width
    if color is None:
        color = np.zeros((H, W), dtype=np.uint32)
    if raw is None:
        raw = np.zeros((H, W, 3), dtype=np.float64)
    if count is None:
        count = np.zeros((H, W), dtype=np.int32)
    sp = numpy_to_spheres(spheres)
    st = Stats()
    rc = L.orc_render(C.byref(cam), sp, len(spheres), color.ctypes.data_as(C.POINTER(C.c_uint32)),
                      raw.ctypes.data_as(C.POINTER(C.c_double)), count.ctypes.data_as(C.POINTER(C.c_int32)),
                      color.size, seed, sample_offset, iterative, row_begin, row_step, threads, C.byref(st))
    if rc != 0:
        raise RuntimeError("orc_render failed: %d" % rc)
    return color, raw, count, st


def camera_ray_entries(cam, px, seed=1, sample_offset=0):
    """CameraRayIterator::next + RayColorEntry::new for (pixel_i, pixel_j, pixel_sample_num) rows of `px`."""
    px = np.ascontiguousarray(px, dtype=np.int32).reshape(-1, 3)
    out = np.zeros(len(px), dtype=PATH_ENTRY_DTYPE)
    lib().orc_camera_ray_entries(C.byref(cam), seed, sample_offset, px.ctypes.data, len(px), out.ctypes.data)
    return out


def ray_color_iteration(entries, spheres, image_width, seed=1, sample_offset=0):
    """One ray_color_iteration on every unfinished entry, in place."""
    sp = numpy_to_spheres(spheres)
    lib().orc_ray_color_iteration_batch(entries.ctypes.data, len(entries), sp, len(spheres), seed, image_width,
                                        sample_offset)
    return entries
