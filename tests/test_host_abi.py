"""CPU-side checks of the product library: it loads, exports every symbol include/dray.h declares, and its
host logic (camera set-up, scene recipes, flattening, partition arithmetic, error reporting) agrees with the
oracle.  No GPU compute is called here; GPU parity lives in the tests marked `gpu`."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import _oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(dray):
    from dray_amd import _ffi
    header = open(os.path.join(ROOT, "include", "dray.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)   # declarations only, not the prose
    declared = set(re.findall(r"\b(dray_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_ffi.EXPORTS), declared ^ set(_ffi.EXPORTS)
    L = _ffi.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert dray.version().startswith("dray-hip")
    assert L.dray_strerror(0) == b"ok" and b"buffer" in L.dray_strerror(-2) and b"unknown" in L.dray_strerror(-99)


def test_struct_layouts_match_header(dray):
    from dray_amd import _ffi
    assert C.sizeof(_ffi.CameraPOD) == C.sizeof(O.Camera) == 8 + 16 + 8 + 72 + 16
    assert C.sizeof(_ffi.SpherePOD) == C.sizeof(O.Sphere) == 72
    assert C.sizeof(_ffi.Opts) == 40 and C.sizeof(_ffi.Stats) == 64
    o = _ffi.default_opts()
    assert (o.seed, o.mode, o.flags, o.sample_offset, o.device, o.row_begin, o.row_step) == (1, 1, 0, 0, -1, 0, 1)


def cameras(dray):
    yield dray.Camera.new()
    yield dray.camera_final()
    yield dray.camera_three_spheres()
    c = dray.camera_final(width=333, spp=7, depth=3)
    c.vfov, c.lookfrom, c.lookat, c.vup = 37.5, (1.5, -2.25, 7.0), (0.3, 0.1, -0.2), (0.1, 1.0, 0.2)
    c.defocus_angle, c.focus_dist, c.aspect_ratio = 2.5, 3.4, 1.37
    yield c


def test_camera_initialize_matches_oracle_bit_for_bit(dray):
    for cam in cameras(dray):
        d = cam.initialize()
        oc = O.default_camera()
        oc.aspect_ratio, oc.image_width, oc.samples_per_pixel, oc.max_depth = cam.aspect_ratio, cam.image_width, cam.samples_per_pixel, cam.max_depth
        oc.vfov, oc.lookfrom, oc.lookat, oc.vup = cam.vfov, O.v3(*cam.lookfrom), O.v3(*cam.lookat), O.v3(*cam.vup)
        oc.defocus_angle, oc.focus_dist = cam.defocus_angle, cam.focus_dist
        od = O.camera_initialize(oc)
        assert d.image_height == od.image_height and d.pixel_samples_scale == od.pixel_samples_scale
        for f in ("center", "pixel00_loc", "pixel_delta_u", "pixel_delta_v", "u", "v", "w", "defocus_disk_u", "defocus_disk_v"):
            assert tuple(getattr(d, f)) == tuple(getattr(od, f)), f
    cam = dray.Camera.new()                               # Camera::new defaults (camera.rs:147-163)
    assert (cam.aspect_ratio, cam.image_width, cam.samples_per_pixel, cam.max_depth, cam.vfov) == (1.0, 100, 10, 10, 90.0)
    assert cam.lookat == (0.0, 0.0, -1.0) and cam.vup == (0.0, 1.0, 0.0) and cam.focus_dist == 10.0
    cam.initialize()
    assert cam.image_height == 100
    bad = dray.Camera.new()
    bad.image_width = 0
    with pytest.raises(dray.DrayError) as e:
        bad.initialize()
    assert e.value.status == -1


def test_scene_recipes_match_oracle_and_fixture(dray):
    for seed in (1, 2, 12345678901234567):
        a, b = dray.scene_final(seed), O.scene_final(seed)
        assert len(a) == len(b)
        for f in ("center", "radius", "kind", "albedo", "fuzz_or_ior"):
            assert (a[f] == b[f]).all(), f
    a, b = dray.scene_three_spheres(), O.scene_three_spheres()
    for f in ("center", "radius", "kind", "albedo", "fuzz_or_ior"):
        assert (a[f] == b[f]).all()
    txt = np.loadtxt(os.path.join(ROOT, "tests", "golden", "scene_final_seed1.txt"))
    sp = dray.scene_final(1)
    assert (txt[:, 0:3] == sp["center"]).all() and (txt[:, 5:8] == sp["albedo"]).all() and (txt[:, 8] == sp["fuzz_or_ior"]).all()
    from dray_amd import _ffi
    small = np.zeros(10, dtype=_ffi.SPHERE_DTYPE)
    assert _ffi.lib().dray_scene_final(1, small.ctypes.data, len(small)) == _ffi.ERR_BUFFER_TOO_SMALL


def test_world_flattening(dray):
    w = dray.HittableList()
    w.add(dray.Sphere((0, -1000, 0), 1000.0, dray.Lambertian((0.5, 0.5, 0.5))))
    w.add(dray.Sphere((1, 2, 3), -4.0, dray.Metal((0.7, 0.6, 0.5), 3.0)))   # radius -> 0, fuzz -> 1
    inner = dray.HittableList([dray.Sphere((0, 1, 0), 1.0, dray.Dialectric(1.5)), dray.Sphere((9, 9, 9), 2.0, dray.DefaultMaterial())])
    w.add(inner)
    flat = dray.flatten_world(w)
    assert w.len() == 3 and len(flat) == 4
    assert list(flat["kind"]) == [1, 2, 3, 0] and flat["radius"][1] == 0.0 and flat["fuzz_or_ior"][1] == 1.0
    assert flat["fuzz_or_ior"][2] == 1.5 and tuple(flat["center"][3]) == (9.0, 9.0, 9.0)
    assert w[0].radius() == 1000.0 and w[0].center() == (0.0, -1000.0, 0.0)

    class BoundingBox(dray.Hittable):     # a hittable without a GPU form is refused, not skipped
        pass

    w.add(BoundingBox())
    with pytest.raises(dray.DrayError) as e:
        dray.flatten_world(w)
    assert e.value.status == -3
    w.clear()
    assert len(dray.flatten_world(w)) == 0


def test_no_gpu_means_loud_failure(dray):
    """Without a device the render call reports an error; nothing falls back to a CPU path."""
    if dray.device_count() > 0:
        pytest.skip("a GPU is present")
    cam = dray.camera_three_spheres(width=16, spp=1, depth=2)
    buf = np.zeros(16 * 9, dtype=np.uint32)
    with pytest.raises(dray.DrayError) as e:
        cam.render(dray.scene_three_spheres(), None, buf, None, None)
    assert e.value.status == -4
    assert (buf == 0).all()
    for devices in ([0], [0, 1], "all"):   # the several-GPU call likewise
        cam.devices = devices
        with pytest.raises(dray.DrayError) as e:
            cam.render(dray.scene_three_spheres(), None, buf, None, None)
        assert e.value.status == -4 and (buf == 0).all()


def test_partition_arithmetic(dray):
    from dray_amd import partition as P
    for H in (1, 2, 7, 54, 675):
        for G in (1, 2, 3, 4, 8):
            rows = []
            for r in range(G):
                b, s, n = P.row_partition(H, G, r)
                mine = list(range(b, H, s))
                assert len(mine) == n <= P.slab_rows(H, G)
                rows += mine
            assert sorted(rows) == list(range(H))
            slabs = np.full((G, P.slab_rows(H, G), 5), -1, dtype=np.int32)
            for r in range(G):
                for k, j in enumerate(range(r, H, G)):
                    slabs[r, k, :] = j
            img = P.deinterleave(slabs, H, G)
            assert img.shape == (H, 5) and (img == np.arange(H)[:, None]).all()
    with pytest.raises(ValueError):
        P.row_partition(10, 2, 2)


def read_ppm(path):
    """(width, height, uint8[H, W, 3]) of a P3 or P6 file with maxval 255."""
    with open(path, "rb") as f:
        data = f.read()
    head = data.split(None, 4)
    magic, w, h, maxval = head[0], int(head[1]), int(head[2]), int(head[3])
    assert maxval == 255
    if magic == b"P6":
        body = data[len(data) - 3 * w * h:]
        px = np.frombuffer(body, dtype=np.uint8)
    else:
        assert magic == b"P3"
        px = np.array(data.split()[4:], dtype=np.int64).astype(np.uint8)
    return w, h, px.reshape(h, w, 3)


def test_write_ppm_plain_and_binary(dray, tmp_path):
    """The image file the reference opens but never writes (main.rs:8,52): P3 one pixel per line, and P6."""
    rng = np.random.default_rng(5)
    color = (rng.integers(0, 1 << 24, size=(5, 7), dtype=np.uint32) | np.uint32(0xFF000000))
    want = np.stack([(color >> 16) & 255, (color >> 8) & 255, color & 255], axis=-1).astype(np.uint8)
    for binary in (0, 1):
        path = str(tmp_path / ("img%d.ppm" % binary))
        dray.write_ppm(path, color, binary=bool(binary))
        w, h, px = read_ppm(path)
        assert (w, h) == (7, 5) and (px == want).all()
    text = open(str(tmp_path / "img0.ppm")).read().splitlines()
    assert text[:3] == ["P3", "7 5", "255"] and len(text) == 3 + 35
    assert text[3] == "%d %d %d" % tuple(want[0, 0])
    with pytest.raises(dray.DrayError):
        dray.write_ppm(str(tmp_path / "no_such_dir" / "x.ppm"), color)
    with pytest.raises(dray.DrayError):
        dray.write_ppm(path, color[:0])


def test_cpp_mirror_builds_and_fails_loudly_without_gpu(dray, tmp_path):
    """The C++ mirror (include/dray.hpp) compiles with the host compiler alone; without a GPU its render throws
    dray::Error carrying DRAY_ERR_NO_DEVICE instead of falling back to anything."""
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    subprocess.run(["make", "-C", os.path.join(here, "cpp"), "--no-print-directory"], check=True,
                   stdout=subprocess.DEVNULL)
    exe = os.path.join(here, "cpp", "mirror_render")
    assert os.path.exists(exe)
    if dray.device_count() > 0:
        pytest.skip("a GPU is visible: the failure path is not reachable")
    p = subprocess.run([exe, "0", "21", "48", "4", "10", str(tmp_path / "x.ppm")], capture_output=True, text=True)
    assert p.returncode == 4 and "no HIP device" in p.stderr
    assert not (tmp_path / "x.ppm").exists()


def _cell_of(info, p):
    n = np.array(list(info.n))
    lo, hi = np.array(list(info.lo)), np.array(list(info.hi))
    c = np.floor((p - lo) / ((hi - lo) / n)).astype(np.int64)
    return c


def test_grid_lists_every_sphere_wherever_its_ball_reaches(dray):
    """The grid scan may skip a sphere only in cells its ball does not reach.  For the reference scene, a 10 k-sphere
    scene and a random cloud: every point of every listed sphere's ball (sampled on and inside the surface, and the
    8 corners of its box shrunk onto the ball) lies in a cell that lists the sphere; loose and wide spheres are
    exactly those the builder reports; every narrow sphere is listed or loose."""
    rng = np.random.default_rng(11)
    cloud = np.zeros(400, dtype=dray._ffi.SPHERE_DTYPE)
    cloud["center"] = rng.uniform(-6, 6, size=(400, 3))
    cloud["radius"] = rng.uniform(0.02, 1.2, size=400)
    cloud["radius"][0] = 9.0     # spans too many cells: loose
    cloud["radius"][1] = 500.0   # wide
    cloud["center"][2] = (np.nan, 0, 0)  # not finite: loose
    cloud["kind"] = 1
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from fuzz_scans import random_scene   # the generator of the GPU fuzz tests: layouts, scales, offsets, outsized spheres
    frng = np.random.default_rng(21)
    fuzz = [random_scene(frng, dray)[0] for _ in range(25)]
    for spheres in [dray.scene_final(1), dray.scene_grid_10k(2)[::5], cloud] + fuzz:
        info, cell_begin, entry, loose = dray.grid_inspect(spheres)
        n = np.array(list(info.n))
        assert (n >= 1).all() and (n <= 256).all()
        assert info.n_narrow + info.n_wide == len(spheres)
        wide = np.flatnonzero(np.maximum(spheres["radius"], 0) >= 32.0)
        assert info.n_wide == len(wide)
        listed = set(entry.tolist())
        assert listed.isdisjoint(loose.tolist()) and listed.isdisjoint(wide.tolist())
        assert len(listed) + len(loose) == info.n_narrow
        assert int(np.diff(cell_begin.astype(np.int64)).max()) == info.max_cell_entries
        cells = {}
        for k in sorted(listed)[:: max(1, len(listed) // 150)]:
            c, r = spheres["center"][k], max(float(spheres["radius"][k]), 0.0)
            dirs = rng.normal(size=(40, 3))
            dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
            corners = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]) / np.sqrt(3.0)
            pts = np.concatenate([c + r * dirs, c + r * rng.uniform(0, 1, size=(40, 1)) * dirs, c + r * corners, c[None]])
            for p in pts:
                cx, cy, cz = _cell_of(info, p)
                assert 0 <= cx < n[0] and 0 <= cy < n[1] and 0 <= cz < n[2], "the grid box holds every listed ball"
                ci = (cz * n[1] + cy) * n[0] + cx
                if ci not in cells:
                    cells[ci] = set(entry[cell_begin[ci]:cell_begin[ci + 1]].tolist())
                assert k in cells[ci], (k, p)
        # a cell's entries are in insertion order
        for ci in list(cells)[:50]:
            e = entry[cell_begin[ci]:cell_begin[ci + 1]]
            assert (np.diff(e) > 0).all()
    info, cell_begin, entry, loose = dray.grid_inspect(cloud)
    assert 0 in loose.tolist() and 2 in loose.tolist()
    info, _, _, _ = dray.grid_inspect(np.zeros(0, dtype=dray._ffi.SPHERE_DTYPE))
    assert list(info.n) == [0, 0, 0] and info.n_entries == 0


def test_entry_points_reject_bad_arguments(dray):
    """Every entry point answers null pointers, non-positive sizes and out-of-range options with a status, never with
    a crash (nothing may unwind or fault across the boundary: the reference aborts on panic, Cargo.toml:32-35)."""
    from dray_amd import _ffi
    L = _ffi.lib()
    INV = _ffi.ERR_INVALID_ARG
    cam, der, opts, st = _ffi.CameraPOD(), _ffi.CameraDerivedPOD(), _ffi.Opts(), _ffi.Stats()
    L.dray_camera_default(C.byref(cam))
    L.dray_opts_default(C.byref(opts))
    assert L.dray_camera_initialize(None, C.byref(der)) == INV and L.dray_camera_initialize(C.byref(cam), None) == INV
    bad = _ffi.CameraPOD.from_buffer_copy(cam)
    for field, value in (("image_width", 0), ("image_width", -5), ("aspect_ratio", 0.0), ("aspect_ratio", float("nan")),
                         ("aspect_ratio", float("inf")), ("aspect_ratio", -1.0)):
        bad = _ffi.CameraPOD.from_buffer_copy(cam)
        setattr(bad, field, value)
        assert L.dray_camera_initialize(C.byref(bad), C.byref(der)) == INV, (field, value)
    color = np.zeros(100 * 100, np.uint32)
    sp = dray.scene_three_spheres()
    n = len(sp)
    # the render calls: argument checks come before any device is touched
    assert L.dray_render(None, sp.ctypes.data, n, color.ctypes.data, None, None, color.size, C.byref(opts), None) == INV
    assert L.dray_render(C.byref(cam), sp.ctypes.data, n, None, None, None, color.size, C.byref(opts), None) == INV
    assert L.dray_render(C.byref(cam), None, n, color.ctypes.data, None, None, color.size, C.byref(opts), None) == INV
    assert L.dray_render_multi(None, sp.ctypes.data, n, color.ctypes.data, None, None, color.size, C.byref(opts), None, 0, None) == INV
    assert L.dray_render_multi(C.byref(cam), None, n, color.ctypes.data, None, None, color.size, C.byref(opts), None, 0, None) == INV
    strided = _ffi.Opts.from_buffer_copy(opts)
    strided.row_step = 2
    assert L.dray_render_multi(C.byref(cam), sp.ctypes.data, n, color.ctypes.data, None, None, color.size, C.byref(strided), None, 0, None) == INV
    handle = C.c_void_p()
    assert L.dray_scene_create(None, 3, -1, C.byref(handle)) in (INV, _ffi.ERR_NO_DEVICE) and not handle.value
    assert L.dray_scene_create(sp.ctypes.data, n, -1, None) == INV
    L.dray_scene_destroy(None)   # a no-op, like free(NULL)
    assert L.dray_render_device(C.byref(cam), None, color.ctypes.data, None, None, color.size, C.byref(opts), None, None) == INV
    ticket = C.c_uint64(7)
    assert L.dray_render_device_async(C.byref(cam), None, color.ctypes.data, None, None, color.size, C.byref(opts), None,
                                      C.byref(ticket)) == INV and ticket.value == 0
    assert L.dray_scene_collect(None, 1, C.byref(_ffi.Stats())) == INV
    assert L.dray_context_render(None, C.byref(cam), color.ctypes.data, None, None, color.size, C.byref(opts), 3, None) == INV
    assert L.dray_context_render_progressive(None, C.byref(cam), color.ctypes.data, None, None, color.size, C.byref(opts), 5,
                                             _ffi.PRESENT_FN(), None, None) == INV
    assert L.dray_camera_rays(None, None, 1, None, None) == INV
    assert L.dray_path_step(None, 10, None, 1, None, None) == INV
    assert L.dray_camera_rays_device(None, None, 1, None, None, None) == INV
    assert L.dray_path_step_device(None, 10, None, 1, None, None) == INV
    info = _ffi.GridInfo()
    assert L.dray_grid_inspect(None, 3, C.byref(info), None, None, None) == INV
    assert L.dray_grid_inspect(sp.ctypes.data, n, None, None, None, None) == INV
    unknown = sp.copy()
    unknown["kind"][1] = 9
    assert L.dray_grid_inspect(unknown.ctypes.data, n, C.byref(info), None, None, None) == _ffi.ERR_UNSUPPORTED
    TOO_SMALL = _ffi.ERR_BUFFER_TOO_SMALL   # no room to write to, a null buffer included
    assert L.dray_scene_three_spheres(None, 4) == TOO_SMALL and L.dray_scene_three_spheres(sp.ctypes.data, 2) == TOO_SMALL
    assert L.dray_scene_final(1, None, 1000) == TOO_SMALL and L.dray_scene_final(1, sp.ctypes.data, n) == TOO_SMALL
    assert L.dray_scene_recipe(1, 5, 4, 0.8, 0.15, 0.5, sp.ctypes.data, n) == INV            # hi < lo
    assert L.dray_scene_recipe(1, 5, 5, 0.8, 0.15, 0.5, sp.ctypes.data, n) == 4              # no cells: ground + the three large spheres
    assert L.dray_write_ppm(None, color.ctypes.data, 10, 10, 0) == INV
    assert L.dray_write_ppm(b"/tmp/x.ppm", None, 10, 10, 0) == INV
    assert L.dray_write_ppm(b"/tmp/x.ppm", color.ctypes.data, 0, 10, 0) == INV
    L.dray_opts_default(None)
    L.dray_camera_default(None)


def test_public_headers_compile_as_c_and_cpp(tmp_path):
    """include/dray.h is what a foreign-function binding reads: it must be plain C (C99, pedantic) as well as C++, and a
    C program must be able to call through it; include/dray.hpp needs nothing newer than C++14."""
    import subprocess
    inc = os.path.join(ROOT, "include")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-x", "c",
                    os.path.join(inc, "dray.h")], check=True)
    subprocess.run(["g++", "-std=c++14", "-Wall", "-Wextra", "-fsyntax-only", "-x", "c++", "-include",
                    os.path.join(inc, "dray.hpp"), "/dev/null"], check=True)
    src = tmp_path / "abi.c"
    src.write_text('#include <stdio.h>\n#include "dray.h"\n'
                   "int main(void) {\n"
                   "    dray_camera cam; dray_camera_derived d; dray_opts o; dray_sphere s[4];\n"
                   "    dray_camera_default(&cam); dray_opts_default(&o);\n"
                   "    cam.aspect_ratio = 16.0 / 9.0; cam.image_width = 400;\n"
                   "    if (dray_camera_initialize(&cam, &d) != DRAY_OK || d.image_height != 225) return 1;\n"
                   "    if (dray_scene_three_spheres(s, 4) != 4 || s[2].kind != DRAY_MAT_DIELECTRIC) return 2;\n"
                   '    printf("%s %d\\n", dray_version(), (int)sizeof(dray_path_entry));\n'
                   "    return 0;\n}\n")
    exe = tmp_path / "abi"
    lib = os.path.join(ROOT, "distributed-raytracer_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", inc, str(src), "-o", str(exe), "-L", lib,
                    "-ldray_hip", "-Wl,-rpath," + lib, "-Wl,-rpath-link,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert out[0].startswith("dray-hip") and out[-1] == "128"
