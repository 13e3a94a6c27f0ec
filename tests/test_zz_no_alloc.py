"""The allocation-free entry point, dray_render_device_noalloc, against dray_render_device through the C ABI.

It renders with the library's static device workspace instead of buffers the scene allocates, so its results must
be those of dray_render_device bit for bit: two calls in a row on one GPU, the second on a smaller frame (the
workspace keeps the first call's contents and everything that must start at zero has to be cleared), then a frame
beyond the workspace, which must be refused with a clear error and leave the buffers alone.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _render(dray, gpu, scene, cam, noalloc, **opts):
    color, raw, count = gpu.alloc_buffers(cam, 0)
    st = gpu.render_device(cam, scene, color, raw, count, noalloc=noalloc, **opts)
    return gpu.color_to_numpy(color).copy(), raw.cpu().numpy().copy(), count.cpu().numpy().copy(), st


def test_noalloc_entry_point_equals_render_device(dray):
    import torch
    from dray_amd import device as gpu
    assert torch.cuda.is_available()
    spheres = dray.scene_final(1)
    scene = gpu.DeviceScene(spheres, 0)
    # first call: a frame of 200 x 112 at 24 samples; second call: a smaller one, fewer samples, another seed
    frames = [(200, 24, 50, 7), (72, 5, 12, 8)]
    for mode in (dray.MODE_F32, dray.MODE_F64):
        for width, spp, depth, seed in frames:
            cam = dray.camera_final(width=width, spp=spp, depth=depth)
            cam.mode, cam.seed = mode, seed
            got = _render(dray, gpu, scene, cam, True)
            want = _render(dray, gpu, scene, cam, False)
            assert (got[0] == want[0]).all(), (mode, width)
            assert (got[1] == want[1]).all() and (got[2] == want[2]).all(), (mode, width)
            assert got[3].segments == want[3].segments and got[3].samples == want[3].samples
            assert got[3].segments > 0 and (got[2] == spp).all()
    # accumulating across two allocation-free calls equals one call of all the samples (colors.rs:38-39)
    cam = dray.camera_final(width=96, spp=6, depth=20)
    cam.mode, cam.seed = dray.MODE_F32, 3
    whole = _render(dray, gpu, scene, cam, False)
    color, raw, count = gpu.alloc_buffers(cam, 0)
    cam.samples_per_pixel = 4
    gpu.render_device(cam, scene, color, raw, count, noalloc=True, sample_offset=0)
    cam.samples_per_pixel = 2
    gpu.render_device(cam, scene, color, raw, count, noalloc=True, sample_offset=4, want_stats=False)
    torch.cuda.synchronize()
    assert (gpu.color_to_numpy(color) == whole[0]).all() and (raw.cpu().numpy() == whole[1]).all()
    assert (count.cpu().numpy() == 6).all()
    # a row partition keeps a large frame inside the workspace; the whole frame does not fit and is refused
    big = dray.camera_final(width=1600, spp=1, depth=2)   # 1600 x 900 = 1 440 000 pixels > 1 048 576
    big.mode = dray.MODE_F32
    color, raw, count = gpu.alloc_buffers(big, 0)
    color.fill_(0x55)
    with pytest.raises(dray.DrayError) as e:
        gpu.render_device(big, scene, color, raw, count, noalloc=True)
    assert e.value.status == dray._ffi.ERR_TOO_LARGE and "workspace" in str(e.value)
    assert (color == 0x55).all() and (count == 0).all()
    half = gpu.render_device(big, scene, color, raw, count, noalloc=True, row_begin=0, row_step=2)
    assert half.rows_rendered == 450 and half.samples == 1600 * 450
    ref_c, ref_r, ref_n = gpu.alloc_buffers(big, 0)
    ref_c.fill_(0x55)
    gpu.render_device(big, scene, ref_c, ref_r, ref_n, row_begin=0, row_step=2)
    assert (color == ref_c).all() and (raw == ref_r).all() and (count == ref_n).all()
    scene.close()


def test_noalloc_symbol_and_limit(dray):
    """The declaration's limit is the one the library enforces (no GPU work here beyond the refusal above)."""
    L = dray._ffi.lib()
    assert hasattr(L, "dray_render_device_noalloc")
    assert dray._ffi.NOALLOC_MAX_PIXELS == 1 << 20
    msg = L.dray_strerror(dray._ffi.ERR_TOO_LARGE)
    assert b"workspace" in msg
