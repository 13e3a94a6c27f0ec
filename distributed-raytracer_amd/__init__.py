"""MI355X (gfx950) backend for the path tracer's per-pixel render path.

The directory name has a hyphen, so import it with
``importlib.import_module("distributed-raytracer_amd")`` or through the
``dray_amd`` shim module at the repository root.
"""
from . import _ffi, partition, raytracer  # noqa: F401
from ._ffi import (FLAG_COMPACT_ROWS, FLAG_ITERATIVE_PRODUCT, FLAG_NO_REFILL, MODE_F32, MODE_F32_PLAIN, MODE_F64, SCAN_AUTO,  # noqa: F401
                   PATH_ENTRY_DTYPE, PIXEL_INDEX_DTYPE, SCAN_GRID, SCAN_HYBRID, SCAN_SCALAR, DrayError)
from .raytracer import (BoundingBox, Camera, DefaultMaterial, Dialectric, Hittable, HittableList, Lambertian, Material, Metal, RenderContext,  # noqa: F401
                        Sphere, camera_final, camera_three_spheres, flatten_world, grid_inspect, scene_final, scene_glass_metal,
                        scene_grid_10k, scene_recipe, scene_three_spheres, write_ppm, Transparent)


def version():
    return _ffi.lib().dray_version().decode()


def device_count():
    return _ffi.lib().dray_device_count()
