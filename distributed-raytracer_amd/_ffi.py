"""ctypes binding of libdray_hip.so (C ABI declared in include/dray.h).

The shared object is built in-tree by ``__graft_entry__.build()`` (or
``make -C distributed-raytracer_amd/csrc``).  There is no CPU fallback: if the
library is missing, loading fails loudly.
"""
import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# DRAY_HIP_LIB selects another build of the same library (A/B timing of two builds on one GPU).
LIB_PATH = os.environ.get("DRAY_HIP_LIB") or os.path.join(PKG_DIR, "libdray_hip.so")

D3 = C.c_double * 3

OK = 0
ERR_INVALID_ARG = -1
ERR_BUFFER_TOO_SMALL = -2
ERR_UNSUPPORTED = -3
ERR_NO_DEVICE = -4
ERR_HIP = -5
ERR_INTERNAL = -6
ERR_TOO_LARGE = -7
NOALLOC_MAX_PIXELS = 1048576
SYNC_UPLOAD_SUMS, SYNC_DOWNLOAD_SUMS = 1, 2

MAT_DEFAULT, MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_TRANSPARENT = 0, 1, 2, 3, 4
MODE_F64, MODE_F32, MODE_F32_PLAIN = 0, 1, 2
SHAPE_SPHERE, SHAPE_BOX = 0, 1
FLAG_ITERATIVE_PRODUCT = 1
FLAG_COMPACT_ROWS = 2
FLAG_NO_REFILL = 4
SCAN_AUTO, SCAN_SCALAR, SCAN_HYBRID, SCAN_GRID = 0, 1, 2, 3


class CameraPOD(C.Structure):
    """dray_camera: the ten public fields of the reference camera (camera.rs:42-53)."""
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


class CameraDerivedPOD(C.Structure):
    """dray_camera_derived: the private fields set by Camera::initialize (camera.rs:55-65)."""
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
    ]


class SpherePOD(C.Structure):
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
assert SPHERE_DTYPE.itemsize == C.sizeof(SpherePOD) == 72


class Opts(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64),
        ("mode", C.c_uint32),
        ("flags", C.c_uint32),
        ("sample_offset", C.c_uint32),
        ("device", C.c_int32),
        ("row_begin", C.c_int32),
        ("row_step", C.c_int32),
        ("pixels_per_pool", C.c_int32),
        ("variant", C.c_int32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("segments", C.c_uint64),
        ("samples", C.c_uint64),
        ("kernel_ms", C.c_double),
        ("resolve_ms", C.c_double),
        ("total_ms", C.c_double),
        ("image_height", C.c_int32),
        ("rows_rendered", C.c_int32),
        ("n_spheres", C.c_int32),
        ("n_wide_spheres", C.c_int32),
        ("walk_steps", C.c_uint64),
    ]


# Every symbol include/dray.h declares; tests check that the library exports all of them.
PIXEL_INDEX_DTYPE = np.dtype([("pixel_i", "<i4"), ("pixel_j", "<i4"), ("pixel_sample_num", "<i4")])
PATH_ENTRY_DTYPE = np.dtype(
    [("attenuation", "<f8", 3), ("origin", "<f8", 3), ("direction", "<f8", 3), ("color", "<f8", 3), ("depth", "<i4"),
     ("stage", "<u4"), ("px", PIXEL_INDEX_DTYPE), ("finished", "<u4"), ("hit_object_or_stop", "<u4"), ("_pad", "<u4")])
assert PATH_ENTRY_DTYPE.itemsize == 128

class GridInfo(C.Structure):
    _fields_ = [("lo", C.c_double * 3), ("hi", C.c_double * 3), ("n", C.c_int32 * 3), ("n_narrow", C.c_int32),
                ("n_wide", C.c_int32), ("n_loose", C.c_int32), ("n_entries", C.c_uint64),
                ("max_cell_entries", C.c_uint32), ("_pad", C.c_uint32)]


PRESENT_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.c_int32)   # dray_present_fn: user, pass, samples done

EXPORTS = (
    "dray_version", "dray_strerror", "dray_device_count", "dray_opts_default", "dray_camera_default",
    "dray_camera_initialize", "dray_render", "dray_render_multi", "dray_scene_create", "dray_scene_destroy", "dray_render_device",
    "dray_scene_final", "dray_scene_recipe", "dray_scene_three_spheres", "dray_write_ppm", "dray_camera_rays",
    "dray_path_step", "dray_camera_rays_device", "dray_path_step_device", "dray_grid_inspect",
    "dray_render_device_noalloc", "dray_render_device_async", "dray_scene_collect", "dray_release_cache", "dray_context_create", "dray_context_destroy",
    "dray_context_set_scene", "dray_context_render", "dray_context_render_progressive", "dray_context_pin", "dray_context_unpin", "dray_context_host_alloc",
)

_lib = None


class DrayError(RuntimeError):
    """A non-zero dray_status; maps to the io::Error the Rust binding would return."""

    def __init__(self, status, message):
        super().__init__("dray status %d: %s" % (status, message))
        self.status = status


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with __graft_entry__.build() or `make -C %s`; there is no CPU fallback"
            % (LIB_PATH, os.path.join(PKG_DIR, "csrc")))
    try:
        # torch ships its own libamdhip64; load it first so that this process has ONE HIP runtime and the
        # streams and device pointers torch hands over belong to the runtime the library calls into
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    P = C.POINTER
    L.dray_version.restype = C.c_char_p
    L.dray_version.argtypes = []
    L.dray_strerror.restype = C.c_char_p
    L.dray_strerror.argtypes = [C.c_int]
    L.dray_device_count.argtypes = []
    L.dray_opts_default.restype = None
    L.dray_opts_default.argtypes = [P(Opts)]
    L.dray_camera_default.restype = None
    L.dray_camera_default.argtypes = [P(CameraPOD)]
    L.dray_camera_initialize.argtypes = [P(CameraPOD), P(CameraDerivedPOD)]
    L.dray_render.argtypes = [P(CameraPOD), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                              P(Opts), P(Stats)]
    L.dray_render_multi.argtypes = [P(CameraPOD), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                    P(Opts), C.c_void_p, C.c_int32, P(Stats)]
    L.dray_scene_create.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, P(C.c_void_p)]
    L.dray_scene_destroy.restype = None
    L.dray_scene_destroy.argtypes = [C.c_void_p]
    L.dray_render_device.argtypes = [P(CameraPOD), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, P(Opts),
                                     C.c_void_p, P(Stats)]
    L.dray_render_device_noalloc.argtypes = L.dray_render_device.argtypes
    L.dray_render_device_async.argtypes = [P(CameraPOD), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, P(Opts),
                                           C.c_void_p, P(C.c_uint64)]
    L.dray_scene_collect.argtypes = [C.c_void_p, C.c_uint64, P(Stats)]
    L.dray_release_cache.restype = None
    L.dray_release_cache.argtypes = []
    L.dray_context_create.argtypes = [C.c_void_p, C.c_int32, P(C.c_void_p)]
    L.dray_context_destroy.restype = None
    L.dray_context_destroy.argtypes = [C.c_void_p]
    L.dray_context_set_scene.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.dray_context_render.argtypes = [C.c_void_p, P(CameraPOD), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, P(Opts),
                                      C.c_uint32, P(Stats)]
    L.dray_context_render_progressive.argtypes = [C.c_void_p, P(CameraPOD), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                                  P(Opts), C.c_int32, PRESENT_FN, C.c_void_p, P(Stats)]
    L.dray_context_pin.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.dray_context_host_alloc.argtypes = [C.c_void_p, C.c_size_t, P(C.c_void_p)]
    L.dray_context_unpin.restype = None
    L.dray_context_unpin.argtypes = [C.c_void_p]
    L.dray_scene_final.argtypes = [C.c_uint64, C.c_void_p, C.c_size_t]
    L.dray_scene_three_spheres.argtypes = [C.c_void_p, C.c_size_t]
    L.dray_write_ppm.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    L.dray_camera_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.dray_path_step.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.dray_camera_rays_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    L.dray_path_step_device.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.dray_grid_inspect.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.dray_scene_recipe.argtypes = [C.c_uint64, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_void_p,
                                    C.c_size_t]
    _lib = L
    return L


def check(status):
    if status != OK:
        raise DrayError(status, lib().dray_strerror(status).decode())


def default_opts(**kw):
    o = Opts()
    lib().dray_opts_default(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise TypeError("unknown option %r" % k)
        setattr(o, k, v)
    return o
