"""Device-resident rendering: scene tables and frame buffers stay in HBM.

Thin wrappers over ``dray_scene_create`` / ``dray_render_device`` (include/dray.h)
that take torch tensors for the buffers.  torch is used for device memory and
streams only.
"""
import ctypes as C

import numpy as np

from . import _ffi
from .raytracer import Camera, flatten_world


class DeviceScene:
    """The sphere list of a HittableList uploaded to one GPU."""

    def __init__(self, world, device=0):
        self.spheres = flatten_world(world)
        self.device = int(device)
        h = C.c_void_p()
        _ffi.check(_ffi.lib().dray_scene_create(self.spheres.ctypes.data if len(self.spheres) else None,
                                                len(self.spheres), self.device, C.byref(h)))
        self._h = h

    @property
    def handle(self):
        if self._h is None:
            raise _ffi.DrayError(_ffi.ERR_INVALID_ARG, "scene already destroyed")
        return self._h

    def __len__(self):
        return len(self.spheres)

    def close(self):
        if self._h is not None:
            _ffi.lib().dray_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _ptr(t, dtype, name):
    import torch
    if t is None:
        return None, 0
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.is_contiguous() and t.dtype == dtype):
        raise _ffi.DrayError(_ffi.ERR_INVALID_ARG, "%s must be a contiguous CUDA tensor of %s" % (name, dtype))
    return t.data_ptr(), t.numel()


def alloc_buffers(cam: Camera, device, rows=None, with_raw=True):
    """Zeroed colour / running-sum / count buffers for ``rows`` scanlines (default: the whole image).

    The colour buffer is int32 holding the reference's u32 ARGB bit pattern (colors.rs:42)."""
    import torch
    cam.initialize()
    n_rows = cam.image_height if rows is None else rows
    dev = torch.device("cuda", device)
    color = torch.zeros((n_rows, cam.image_width), dtype=torch.int32, device=dev)
    raw = torch.zeros((n_rows, cam.image_width, 3), dtype=torch.float64, device=dev) if with_raw else None
    count = torch.zeros((n_rows, cam.image_width), dtype=torch.int32, device=dev) if with_raw else None
    return color, raw, count


class Ticket:
    """A render that has been enqueued and not looked at yet (dray_render_device_async); ``stats()`` waits for it."""

    def __init__(self, scene, number):
        self.scene, self.number = scene, int(number)

    def stats(self):
        st = _ffi.Stats()
        if self.number:
            _ffi.check(_ffi.lib().dray_scene_collect(self.scene.handle, C.c_uint64(self.number), C.byref(st)))
        return st


def render_device(cam: Camera, scene: DeviceScene, color, raw=None, count=None, stream=None, want_stats=True,
                  noalloc=False, **opt_overrides):
    """Camera::render with device buffers.  Returns dray_stats; with ``want_stats=False`` None (the call only enqueues
    work on ``stream``); with ``want_stats="later"`` a Ticket whose ``stats()`` waits for the call -- so that several
    renders can be enqueued back to back.  ``noalloc`` takes the entry point that never allocates device memory
    (dray_render_device_noalloc: the library's static workspace, frames of at most NOALLOC_MAX_PIXELS pixels)."""
    import torch
    pc, nc = _ptr(color, torch.int32, "color")
    pr, nr = _ptr(raw, torch.float64, "raw")
    pn, nn = _ptr(count, torch.int32, "count")
    buf_len = nc
    if raw is not None:
        buf_len = min(buf_len, nr // 3)
    if count is not None:
        buf_len = min(buf_len, nn)
    if color.device.index != scene.device:
        raise _ffi.DrayError(_ffi.ERR_INVALID_ARG, "buffers and scene live on different devices")
    if stream is None:
        stream = torch.cuda.current_stream(color.device)
    pod = cam.pod()
    opts = cam.opts(**opt_overrides)
    opts.device = scene.device
    if want_stats == "later" and not noalloc:
        number = C.c_uint64(0)
        _ffi.check(_ffi.lib().dray_render_device_async(C.byref(pod), scene.handle, pc, pr, pn, buf_len, C.byref(opts),
                                                       C.c_void_p(stream.cuda_stream), C.byref(number)))
        return Ticket(scene, number.value)
    st = _ffi.Stats() if want_stats else None
    call = _ffi.lib().dray_render_device_noalloc if noalloc else _ffi.lib().dray_render_device
    rc = call(C.byref(pod), scene.handle, pc, pr, pn, buf_len, C.byref(opts), C.c_void_p(stream.cuda_stream),
              C.byref(st) if want_stats else None)
    _ffi.check(rc)
    cam.last_stats = st
    return st


def color_to_numpy(color):
    """int32 device colour buffer -> uint32 numpy array with the reference's ARGB packing."""
    return color.detach().cpu().numpy().view(np.uint32)


# ------------------------------------------------------- resumable path state --

def camera_rays(cam: Camera, pixel_indices, **opt_overrides):
    """CameraRayIterator::next (camera.rs:103-114) for each (pixel_i, pixel_j, pixel_sample_num) row, in the given
    order, followed by RayColorEntry::new(ray, max_depth) (camera.rs:23-31).  Returns a numpy array of
    PATH_ENTRY_DTYPE records (the reference walks the pixels in a shuffled order, camera.rs:96-98: shuffle the rows)."""
    px = np.ascontiguousarray(pixel_indices, dtype=np.int32).reshape(-1, 3)
    out = np.zeros(len(px), dtype=_ffi.PATH_ENTRY_DTYPE)
    pod = cam.pod()
    opts = cam.opts(**opt_overrides)
    _ffi.check(_ffi.lib().dray_camera_rays(C.addressof(pod), px.ctypes.data, len(px), out.ctypes.data,
                                           C.addressof(opts)))
    return out


def iterate_rays(cam: Camera, shuffle_seed=0, **opt_overrides):
    """Camera::iterate_rays (camera.rs:165, CameraRayIterator :68-116): every pixel of the image once, with sample
    number 0 (the iterator stops after image_width * image_height items, :93-95), in a shuffled order (:96-98;
    the reference shuffles with an OS-seeded generator, here the order is numpy's permutation for ``shuffle_seed``).
    Returns (pixel_indices int32[n, 3], entries): the i-th item of the reference's iterator is
    (PixelIndexEntry(*pixel_indices[i]), entries[i]'s origin and direction), already wrapped as
    RayColorEntry::new(ray, max_depth) the way its consumer does (ray_server.rs:48-51)."""
    cam.initialize()
    W, H = int(cam.image_width), int(cam.image_height)
    order = np.random.default_rng(shuffle_seed).permutation(W * H)
    px = np.stack([order % W, order // W, np.zeros_like(order)], axis=1).astype(np.int32)  # camera.rs:103-106
    return px, camera_rays(cam, px, **opt_overrides)


def ray_color_iteration(entries, scene: DeviceScene, cam: Camera, **opt_overrides):
    """One ray_color_iteration (camera.rs:118-144) on every unfinished entry of the batch, in place; returns the
    call's statistics (segments = entries that were intersected)."""
    if not (isinstance(entries, np.ndarray) and entries.dtype == _ffi.PATH_ENTRY_DTYPE and entries.flags.c_contiguous):
        raise _ffi.DrayError(_ffi.ERR_INVALID_ARG, "entries must be a contiguous PATH_ENTRY_DTYPE array")
    opts = cam.opts(**opt_overrides)
    st = _ffi.Stats()
    _ffi.check(_ffi.lib().dray_path_step(scene.handle, int(cam.image_width), entries.ctypes.data, len(entries),
                                         C.addressof(opts), C.addressof(st)))
    return st
