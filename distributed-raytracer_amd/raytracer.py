"""Host-side mirror of the reference's scene / camera API for the render path.

Names, argument meaning and error behaviour follow ``src/raytracer`` of the
reference (Rust): ``Sphere::new`` (sphere.rs:13), ``HittableList`` (hittable_list.rs:11-40),
``Lambertian/Metal/Dialectric::new`` (material.rs:39,66,89), ``Camera`` with its ten
public fields (camera.rs:41-54), ``Camera::new`` (:147) and ``Camera::render`` (:208).
Only ``render`` does work: it flattens the scene and calls the HIP library through
its C ABI.  The Rust reference cannot be compiled in this environment, so this
Python mirror (and the C++ one in include/dray.hpp) is what the tests drive.
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import (FLAG_COMPACT_ROWS, FLAG_ITERATIVE_PRODUCT, MAT_DEFAULT, MAT_DIELECTRIC, MAT_LAMBERTIAN, MAT_METAL, MAT_TRANSPARENT,
                   MODE_F32, MODE_F32_PLAIN, MODE_F64, DrayError)


def _v3(v):
    v = tuple(float(x) for x in v)
    if len(v) != 3:
        raise ValueError("expected 3 components")
    return v


# ------------------------------------------------------------------ materials --

class Material:
    kind = MAT_DEFAULT
    albedo = (0.0, 0.0, 0.0)
    param = 0.0


class DefaultMaterial(Material):
    """Absorbs every ray (material.rs:10-19)."""


class Transparent(Material):
    """Scatters the incoming ray unchanged with attenuation (1, 1, 1) (material.rs:22-31)."""
    kind = MAT_TRANSPARENT


class Lambertian(Material):
    kind = MAT_LAMBERTIAN

    def __init__(self, albedo):
        self.albedo = _v3(albedo)


class Metal(Material):
    kind = MAT_METAL

    def __init__(self, albedo, fuzz):
        self.albedo = _v3(albedo)
        self.param = float(fuzz) if fuzz < 1.0 else 1.0  # material.rs:67


class Dialectric(Material):  # spelling as in the reference (material.rs:84)
    kind = MAT_DIELECTRIC

    def __init__(self, refraction_index):
        self.param = float(refraction_index)


# ------------------------------------------------------------------- hittables --

class Hittable:
    """Anything that can be put into a HittableList (hittable.rs:25-28)."""

    def flatten(self, out):
        raise DrayError(_ffi.ERR_UNSUPPORTED, "%s has no GPU representation" % type(self).__name__)


class Sphere(Hittable):
    def __init__(self, center, radius, mat):
        self._center = _v3(center)
        self._radius = max(float(radius), 0.0)  # sphere.rs:14
        self._mat = mat

    def center(self):
        return self._center

    def radius(self):
        return self._radius

    def flatten(self, out):
        if not isinstance(self._mat, Material):
            raise DrayError(_ffi.ERR_UNSUPPORTED, "unknown material %r" % (self._mat,))
        out.append((self._center, self._radius, self._mat.kind, 0, self._mat.albedo, self._mat.param))


class BoundingBox(Hittable):
    """bounding_box.rs:6-28: three intervals.  As a hittable it is intersected the way bounding_box.rs:50-76 is
    written (axes 0 and 1 only, plane distance ``min - origin / direction``) and its material is always Transparent."""

    def __init__(self, axes=None):
        inf = float("inf")
        axes = [(inf, -inf)] * 3 if axes is None else [(float(lo), float(hi)) for lo, hi in axes]  # default: EMPTY
        if len(axes) != 3:
            raise ValueError("expected 3 intervals")
        self.axes = axes

    @classmethod
    def new_xyz(cls, x_min, x_max, y_min, y_max, z_min, z_max):
        return cls([(x_min, x_max), (y_min, y_max), (z_min, z_max)])

    def flatten(self, out):
        lo = tuple(a[0] for a in self.axes)
        hi = tuple(a[1] for a in self.axes)
        # a dray_box record in a dray_sphere's place: min in `center`, max in `albedo`, shape = box
        out.append((lo, 0.0, 0, _ffi.SHAPE_BOX, hi, 0.0))


class HittableList(Hittable):
    def __init__(self, objects=None):
        self.objects = list(objects) if objects is not None else []

    def add(self, obj):
        self.objects.append(obj)

    def clear(self):
        self.objects.clear()

    def len(self):
        return len(self.objects)

    def __len__(self):
        return len(self.objects)

    def __getitem__(self, i):
        return self.objects[i]

    def flatten(self, out):
        for o in self.objects:  # insertion order is the tie-break order (hittable_list.rs:65-70)
            if not isinstance(o, Hittable):
                raise DrayError(_ffi.ERR_UNSUPPORTED, "unknown hittable %r" % (o,))
            o.flatten(out)


def flatten_world(world):
    """World -> numpy array of dray_sphere records, or pass an already flat array through."""
    if isinstance(world, np.ndarray):
        return np.ascontiguousarray(world, dtype=_ffi.SPHERE_DTYPE)
    rows = []
    world.flatten(rows)
    return np.array(rows, dtype=_ffi.SPHERE_DTYPE) if rows else np.zeros(0, dtype=_ffi.SPHERE_DTYPE)


# --------------------------------------------------------------------- context --

class RenderContext:
    """A dray_context: scenes, device buffers and staging memory kept between render calls (include/dray.h).

    ``devices``: None = the current device, "all" = every visible GPU, or a list of device numbers (share k renders
    scanlines k, k + G, ...).  Give it to a camera (``cam.context = ctx``) and ``Camera.render`` goes through it."""

    def __init__(self, devices=None):
        h = C.c_void_p()
        if devices is None:
            import torch
            devs = np.array([torch.cuda.current_device() if torch.cuda.is_available() else 0], dtype=np.int32)
        elif isinstance(devices, str):
            devs = None
        else:
            devs = np.ascontiguousarray(devices, dtype=np.int32)
        _ffi.check(_ffi.lib().dray_context_create(devs.ctypes.data if devs is not None else None,
                                                  len(devs) if devs is not None else 0, C.byref(h)))
        self._h = h
        self._pinned = []

    @property
    def handle(self):
        if self._h is None:
            raise DrayError(_ffi.ERR_INVALID_ARG, "context already destroyed")
        return self._h

    def set_scene(self, world):
        spheres = flatten_world(world)
        _ffi.check(_ffi.lib().dray_context_set_scene(self.handle, spheres.ctypes.data if len(spheres) else None, len(spheres)))

    def pin(self, *buffers):
        """Register numpy buffers with the HIP runtime so that copies go straight to and from them; the context keeps
        a reference to each until unpin() or close()."""
        for b in buffers:
            if b is None:
                continue
            _ffi.check(_ffi.lib().dray_context_pin(self.handle, b.ctypes.data, b.nbytes))
            self._pinned.append(b)

    def alloc_buffers(self, n_pixels):
        """(color uint32[n], raw float64[n, 3], count int32[n]) in page-locked memory of the context: zeroed, reached by
        the copy engines without staging, valid until close()."""
        out = []
        for dtype, shape in ((np.uint32, (n_pixels,)), (np.float64, (n_pixels, 3)), (np.int32, (n_pixels,))):
            nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
            p = C.c_void_p()
            _ffi.check(_ffi.lib().dray_context_host_alloc(self.handle, nbytes, C.byref(p)))
            buf = (C.c_char * nbytes).from_address(p.value)
            out.append(np.frombuffer(buf, dtype=dtype).reshape(shape))
        return tuple(out)

    def unpin(self):
        if self._h is not None:
            _ffi.lib().dray_context_unpin(self._h)
        self._pinned = []

    def render(self, pod, color_buffer, raw_buffer, count_buffer, buf_len, opts, sync, stats):
        return _ffi.lib().dray_context_render(
            self.handle, C.byref(pod), color_buffer.ctypes.data,
            raw_buffer.ctypes.data if raw_buffer is not None else None,
            count_buffer.ctypes.data if count_buffer is not None else None, buf_len, C.byref(opts), sync, C.byref(stats))

    def render_progressive(self, pod, color_buffer, raw_buffer, count_buffer, buf_len, opts, samples_per_pass, present, stats):
        """dray_context_render_progressive: ``present(pass_index, samples_done)`` (or None) is called after every pass."""
        failure = []

        def trampoline(_user, index, done):   # an exception must not unwind through the C frames
            try:
                if present is not None and not failure:
                    present(index, done)
            except BaseException as e:   # noqa: BLE001 -- re-raised below
                failure.append(e)

        cb = _ffi.PRESENT_FN(trampoline)
        rc = _ffi.lib().dray_context_render_progressive(
            self.handle, C.byref(pod), color_buffer.ctypes.data,
            raw_buffer.ctypes.data if raw_buffer is not None else None,
            count_buffer.ctypes.data if count_buffer is not None else None, buf_len, C.byref(opts), int(samples_per_pass),
            cb, None, C.byref(stats))
        if failure:
            raise failure[0]
        return rc

    def close(self):
        if self._h is not None:
            _ffi.lib().dray_context_destroy(self._h)
            self._h = None
            self._pinned = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------- camera --

class Camera:
    """Public fields and defaults of the reference camera (camera.rs:41-54,147-163)."""

    def __init__(self):
        self.aspect_ratio = 1.0
        self.image_width = 100
        self.samples_per_pixel = 10
        self.max_depth = 10
        self.vfov = 90.0
        self.lookfrom = (0.0, 0.0, 0.0)
        self.lookat = (0.0, 0.0, -1.0)
        self.vup = (0.0, 1.0, 0.0)
        self.defocus_angle = 0.0
        self.focus_dist = 10.0
        # backend options (not in the reference): see dray_opts in include/dray.h
        self.seed = 1
        self.mode = MODE_F32
        self.flags = 0
        self.sample_offset = 0
        self.device = -1
        # > 0: render in passes of this many samples per pixel and refresh the window after each pass (the reference
        # refreshes it after every pixel sample, colors.rs:44); the buffers end up the same as after one pass
        self.samples_per_pass = 0
        # GPUs of this node to spread a render over from this one process: None = the one device above; a list of
        # device numbers (a device may appear more than once); "all" = every visible GPU
        self.devices = None
        # a RenderContext to render through (its devices then count, not `devices`): the world is uploaded and its grid
        # built when it changes, not per call, and the passes of a progressive render leave the sums on the device
        self.context = None
        self.last_stats = None

    @classmethod
    def new(cls):
        return cls()

    def pod(self):
        c = _ffi.CameraPOD()
        c.aspect_ratio = float(self.aspect_ratio)
        c.image_width = int(self.image_width)
        c.samples_per_pixel = int(self.samples_per_pixel)
        c.max_depth = int(self.max_depth)
        c.vfov = float(self.vfov)
        c.lookfrom = _ffi.D3(*_v3(self.lookfrom))
        c.lookat = _ffi.D3(*_v3(self.lookat))
        c.vup = _ffi.D3(*_v3(self.vup))
        c.defocus_angle = float(self.defocus_angle)
        c.focus_dist = float(self.focus_dist)
        return c

    def initialize(self):
        """Camera::initialize (camera.rs:169-206), computed by the library on the host."""
        d = _ffi.CameraDerivedPOD()
        pod = self.pod()
        _ffi.check(_ffi.lib().dray_camera_initialize(C.byref(pod), C.byref(d)))
        self.image_height = d.image_height
        self.pixel_samples_scale = d.pixel_samples_scale
        self.derived = d
        return d

    def opts(self, **kw):
        base = dict(seed=self.seed, mode=self.mode, flags=self.flags, sample_offset=self.sample_offset,
                    device=self.device)
        base.update(kw)
        return _ffi.default_opts(**base)

    def render(self, world, window, color_buffer, raw_buffer, count_buffer, **opt_overrides):
        """Camera::render (camera.rs:208-236) on the GPU.

        color_buffer: uint32[>= W*H]; raw_buffer: float64[>= W*H, 3] or None; count_buffer: int32[>= W*H] or None.
        The buffers are accumulated into, never cleared (colors.rs:38-39).  ``window`` (may be None) gets one
        ``update_with_buffer(color_buffer, W, H)`` per pass (one pass unless ``samples_per_pass`` is set) instead
        of one per pixel sample (colors.rs:44).  Raises DrayError where the Rust binding would return an io::Error.
        """
        spheres = flatten_world(world)
        self.initialize()
        for name, buf, dt in (("color_buffer", color_buffer, np.uint32), ("raw_buffer", raw_buffer, np.float64),
                              ("count_buffer", count_buffer, np.int32)):
            if buf is None:
                if name == "color_buffer":
                    raise DrayError(_ffi.ERR_INVALID_ARG, "color_buffer is required")
                continue
            if not (isinstance(buf, np.ndarray) and buf.dtype == dt and buf.flags.c_contiguous and buf.flags.writeable):
                raise DrayError(_ffi.ERR_INVALID_ARG, "%s must be a writable contiguous %s array" % (name, dt.__name__))
        buf_len = color_buffer.size
        if raw_buffer is not None and raw_buffer.size // 3 < buf_len:
            buf_len = raw_buffer.size // 3
        if count_buffer is not None and count_buffer.size < buf_len:
            buf_len = count_buffer.size
        pod = self.pod()
        opts = self.opts(**opt_overrides)
        total = int(self.samples_per_pixel)
        per_pass = int(self.samples_per_pass) if 0 < int(self.samples_per_pass) < total else total
        first_sample = opts.sample_offset
        done = 0
        whole = None
        ctx = self.context
        if ctx is not None:
            if opts_row_partition(opts):
                raise DrayError(_ffi.ERR_INVALID_ARG, "a context makes the row partition itself")
            ctx.set_scene(spheres)   # a no-op when the list is the one the context holds
            if per_pass < total:
                # the passes of one call, two under way at a time: the sums go up before the first and come back after
                # the last, in between only the colours move, and the window is refreshed while the next pass renders
                st = _ffi.Stats()
                W, H = self.image_width, self.image_height

                def show(_index, _done):
                    if window is not None:
                        window.update_with_buffer(color_buffer, W, H)

                _ffi.check(ctx.render_progressive(pod, color_buffer, raw_buffer, count_buffer, buf_len, opts, per_pass, show, st))
                self.last_stats = st
                return None
        while True:  # at least one call, so that a camera of 0 samples fails (or not) the way one call does
            pod.samples_per_pixel = min(per_pass, total - done) if total > 0 else total
            opts.sample_offset = first_sample + done
            st = _ffi.Stats()
            args = (C.byref(pod), spheres.ctypes.data if len(spheres) else None, len(spheres), color_buffer.ctypes.data,
                    raw_buffer.ctypes.data if raw_buffer is not None else None,
                    count_buffer.ctypes.data if count_buffer is not None else None, buf_len, C.byref(opts))
            if ctx is not None:
                # the sums go up with the first pass and come back with the last; in between only the colours move
                last = total <= 0 or done + pod.samples_per_pixel >= total
                sync = (_ffi.SYNC_UPLOAD_SUMS if done == 0 else 0) | (_ffi.SYNC_DOWNLOAD_SUMS if last else 0)
                rc = ctx.render(pod, color_buffer, raw_buffer, count_buffer, buf_len, opts, sync, st)
            elif self.devices is None:
                rc = _ffi.lib().dray_render(*args, C.byref(st))
            elif isinstance(self.devices, str):  # "all"
                rc = _ffi.lib().dray_render_multi(*args, None, 0, C.byref(st))
            else:
                devs = np.ascontiguousarray(self.devices, dtype=np.int32)
                rc = _ffi.lib().dray_render_multi(*args, devs.ctypes.data, len(devs), C.byref(st))
            _ffi.check(rc)
            if whole is None:
                whole = st
            else:  # the call's totals over its passes
                for f in ("segments", "samples", "walk_steps"):
                    setattr(whole, f, getattr(whole, f) + getattr(st, f))
                for f in ("kernel_ms", "resolve_ms", "total_ms"):
                    if hasattr(whole, f):
                        setattr(whole, f, getattr(whole, f) + getattr(st, f))
            if window is not None:
                window.update_with_buffer(color_buffer, self.image_width, self.image_height)
            done += pod.samples_per_pixel
            if done >= total:
                break
        self.last_stats = whole
        return None  # Ok(())


def opts_row_partition(opts):
    return opts.row_begin != 0 or opts.row_step not in (0, 1) or (opts.flags & FLAG_COMPACT_ROWS) != 0


def write_ppm(path, color_buffer, width=None, height=None, binary=False):
    """Write a colour buffer (uint32 0xFFrrggbb, [H, W] or flat with width/height given) as PPM: the image file
    the reference's local binary opens but never writes (main.rs:8,52)."""
    buf = np.ascontiguousarray(color_buffer, dtype=np.uint32)
    if width is None or height is None:
        if buf.ndim != 2:
            raise DrayError(_ffi.ERR_INVALID_ARG, "width and height are required for a flat buffer")
        height, width = buf.shape
    if buf.size < int(width) * int(height):
        raise DrayError(_ffi.ERR_BUFFER_TOO_SMALL, "colour buffer smaller than width * height")
    _ffi.check(_ffi.lib().dray_write_ppm(str(path).encode(), buf.ctypes.data, int(width), int(height), int(bool(binary))))


def grid_inspect(world):
    """The uniform grid the library builds for ``world`` (host only): (info, cell_begin, entry_sphere, loose_sphere)."""
    spheres = flatten_world(world)
    info = _ffi.GridInfo()
    ptr = spheres.ctypes.data if len(spheres) else None
    _ffi.check(_ffi.lib().dray_grid_inspect(ptr, len(spheres), C.byref(info), None, None, None))
    n_cells = info.n[0] * info.n[1] * info.n[2]
    cell_begin = np.zeros(n_cells + 1, dtype=np.uint32)
    entry = np.zeros(max(int(info.n_entries), 1), dtype=np.int32)
    loose = np.zeros(max(info.n_loose, 1), dtype=np.int32)
    _ffi.check(_ffi.lib().dray_grid_inspect(ptr, len(spheres), C.byref(info), cell_begin.ctypes.data, entry.ctypes.data,
                                            loose.ctypes.data))
    return info, cell_begin, entry[:int(info.n_entries)], loose[:info.n_loose]


# --------------------------------------------------------------- scene recipes --

def scene_final(seed=1):
    """Sphere list of the reference's local binary (main.rs:13-50), uniforms from the Philox stream of ``seed``."""
    arr = np.zeros(600, dtype=_ffi.SPHERE_DTYPE)
    n = _ffi.lib().dray_scene_final(seed, arr.ctypes.data, len(arr))
    if n < 0:
        _ffi.check(n)
    return arr[:n].copy()


def scene_recipe(seed, lo, hi, p_diffuse, p_metal, fuzz_max):
    """The recipe of main.rs:13-50 over cells [lo, hi)^2 with other material shares."""
    arr = np.zeros((hi - lo) * (hi - lo) + 8, dtype=_ffi.SPHERE_DTYPE)
    n = _ffi.lib().dray_scene_recipe(seed, lo, hi, p_diffuse, p_metal, fuzz_max, arr.ctypes.data, len(arr))
    if n < 0:
        _ffi.check(n)
    return arr[:n].copy()


def scene_grid_10k(seed=1):
    """About 10 000 small spheres on a 100 x 100 grid with the reference scene's density and material mix."""
    return scene_recipe(seed, -50, 50, 0.8, 0.15, 0.5)


def scene_glass_metal(seed=1):
    """The reference scene's layout with half glass and half low-fuzz metal: long specular paths."""
    return scene_recipe(seed, -11, 11, 0.0, 0.5, 0.1)


def scene_three_spheres():
    arr = np.zeros(4, dtype=_ffi.SPHERE_DTYPE)
    n = _ffi.lib().dray_scene_three_spheres(arr.ctypes.data, len(arr))
    if n < 0:
        _ffi.check(n)
    return arr[:n].copy()


def camera_final(width=1200, spp=500, depth=50):
    """Camera of the reference's local binary (main.rs:55-66)."""
    cam = Camera()
    cam.aspect_ratio = 16.0 / 9.0
    cam.image_width = width
    cam.samples_per_pixel = spp
    cam.max_depth = depth
    cam.vfov = 20.0
    cam.lookfrom = (13.0, 2.0, 3.0)
    cam.lookat = (0.0, 0.0, 0.0)
    cam.vup = (0.0, 1.0, 0.0)
    cam.defocus_angle = 0.6
    cam.focus_dist = 10.0
    return cam


def camera_three_spheres(width=400, spp=10, depth=10):
    cam = Camera()
    cam.aspect_ratio = 16.0 / 9.0
    cam.image_width = width
    cam.samples_per_pixel = spp
    cam.max_depth = depth
    return cam
