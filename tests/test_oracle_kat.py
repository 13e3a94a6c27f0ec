"""Pins the CPU oracle: the reference's own Vec3 test vectors (src/raytracer/vec3.rs:260-450), the
published Philox4x32-10 known answers, and analytic known answers derived from the cited reference
formulas for everything the reference's tests do not cover (sphere hit, scatter, camera, colour).
The render path beyond Vec3 stays "parity unpinned": the Rust reference cannot be run here."""
import ctypes as C
import math
import os

import numpy as np
import pytest

import _oracle as O

INF = float("inf")


@pytest.fixture(scope="module")
def L():
    return O.lib()


# ------------------------------------------------ the reference's 24 Vec3 tests --

def test_vec3_reference_vectors(L):
    v = O.v3(1.0, 2.0, 3.0)                                        # test_new / test_new_xyz / test_index
    assert (v[0], v[1], v[2]) == (1.0, 2.0, 3.0)
    assert O.vec_op("neg", (1.0, -2.0, 3.0)) == (-1.0, 2.0, -3.0)   # test_neg
    assert O.vec_op("add", (1, 2, 3), (4, 5, 6)) == (5.0, 7.0, 9.0)  # test_add / test_add_assign
    assert O.vec_op("sub", (4, 5, 6), (1, 2, 3)) == (3.0, 3.0, 3.0)  # test_sub / test_sub_assign
    assert O.vec_op("mul", (1, 2, 3), (4, 5, 6)) == (4.0, 10.0, 18.0)  # test_mul_vec
    assert O.vec_op("scale", (1, 2, 3), s=3.0) == (3.0, 6.0, 9.0)   # test_mul_scalar_before/after, test_mul_assign
    assert O.vec_op("div", (3, 6, 9), s=3.0) == (1.0, 2.0, 3.0)     # test_div_scalar / test_div_assign_scalar
    assert O.vec_op("divv", (1, 2, 3), (4, 5, 6)) == (0.25, 0.4, 0.5)  # test_div_vector
    assert O.vec_op("min", (1, 5, 3), (4, 2, 6)) == (1.0, 2.0, 3.0)  # test_min_vec
    assert O.vec_op("max", (1, 5, 3), (4, 2, 6)) == (4.0, 5.0, 6.0)  # test_max_vec
    assert L.orc_v_length_squared(O.v3(1, 2, 3)) == 14.0            # test_length_squared
    assert L.orc_v_length(O.v3(1, 2, 3)) == math.sqrt(14.0)         # test_length
    assert L.orc_v_dot(O.v3(1, 2, 3), O.v3(4, 5, 6)) == 32.0        # test_dot_product
    assert O.vec_op("cross", (1, 2, 3), (4, 5, 6)) == (-3.0, 6.0, -3.0)  # test_cross_product
    # test_unit_vector: exact only when each component is DIVIDED by the length (3/5 = 0.6 but 3*(1/5) != 0.6)
    assert O.vec_op("unit", (0.0, 3.0, 4.0)) == (0.0, 0.6, 0.8)
    assert 3.0 * (1.0 / 5.0) != 0.6
    # test_index_mut / test_display concern container behaviour, not arithmetic
    assert "%s %s %s" % tuple(repr(x) for x in (1.1, 2.2, 3.3)) == "1.1 2.2 3.3"


# ------------------------------------------------------------------- Philox --

def test_philox_known_answers(L):
    def ph(ctr, key):
        o = (C.c_uint32 * 4)()
        L.orc_philox4x32_10((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), o)
        return list(o)
    # Random123 kat_vectors, philox4x32 10 rounds
    assert ph([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert ph([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert ph([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_uniform_mapping(L):
    assert L.orc_u01(0) == 0.0
    assert L.orc_u01(0xFFFFFFFF) == 1.0 - 2.0 ** -24   # half-open like rand::random (prelude.rs:24)
    assert L.orc_u01(0x80000000) == 0.5
    assert np.float32(L.orc_u01(0x12345678)) == L.orc_u01(0x12345678)  # exact in f32 too
    r = O.philox_rng(5, 17, 3, 2)
    a = [L.orc_rng_draw(C.byref(r), k) for k in range(8)]
    o = (C.c_uint32 * 4)()
    L.orc_philox4x32_10((C.c_uint32 * 4)(17, 3, 2, 1), (C.c_uint32 * 2)(5, 0), o)
    assert a[4:] == [L.orc_u01(w) for w in o]        # draw n = word n%4 of block n/4


# --------------------------------------------------------------- sphere hit --

def hit(L, s, ro, rd, tmin=0.001, tmax=INF):
    rec = O.Hit()
    ok = L.orc_sphere_hit(C.byref(s), O.v3(*ro), O.v3(*rd), tmin, tmax, C.byref(rec))
    return ok, rec


def test_sphere_hit_known_answers(L):
    s = O.make_sphere((0, 0, -1), 0.5, O.MAT_LAMBERTIAN, (0.5, 0.5, 0.5))
    ok, rec = hit(L, s, (0, 0, 0), (0, 0, -1))
    assert ok and rec.t == 0.5 and tuple(rec.p) == (0, 0, -0.5) and tuple(rec.normal) == (0, 0, 1) and rec.front_face == 1
    ok, rec = hit(L, s, (0, 0, -1), (0, 0, -1))                   # from the centre: far root, flipped normal
    assert ok and rec.t == 0.5 and tuple(rec.p) == (0, 0, -1.5) and tuple(rec.normal) == (0, 0, 1) and rec.front_face == 0
    assert not hit(L, s, (0, 1, 0), (0, 0, -1))[0]                # line misses: discriminant < 0
    assert not hit(L, s, (0, 0, 0), (0, 0, 1))[0]                 # sphere behind the ray
    ok, rec = hit(L, s, (0, 0, 0), (0, 0, -2))                    # direction is not normalised: t halves
    assert ok and rec.t == 0.25 and tuple(rec.p) == (0, 0, -0.5)
    # strict interval (interval.rs:28-30): a root equal to either bound is rejected
    assert not hit(L, s, (0, 0, 0), (0, 0, -1), tmin=0.5, tmax=1.5)[0]   # near root == tmin, far root == tmax
    ok, rec = hit(L, s, (0, 0, 0), (0, 0, -1), tmin=0.5, tmax=INF)   # near root == tmin -> far root
    assert ok and rec.t == 1.5
    assert not hit(L, s, (0, 0, 0), (0, 0, -1), tmin=0.001, tmax=0.5)[0]  # near root == tmax -> far root >= tmax -> miss
    # a self-intersection below t = 0.001 is skipped
    ok, rec = hit(L, s, (0, 0, -0.5 + 1e-4), (0, 0, 1))
    assert not ok
    # radius is clamped at 0 on construction (sphere.rs:14)
    assert O.make_sphere((0, 0, 0), -3.0, O.MAT_LAMBERTIAN).radius == 0.0


def test_list_hit_order_and_ties(L):
    a = O.make_sphere((0, 0, -2), 0.5, O.MAT_LAMBERTIAN, (1, 0, 0))
    b = O.make_sphere((0, 0, -2), 0.5, O.MAT_LAMBERTIAN, (0, 1, 0))   # same sphere again: equal t
    c = O.make_sphere((0, 0, -1), 0.25, O.MAT_LAMBERTIAN, (0, 0, 1))  # closer
    rec = O.Hit()
    arr = (O.Sphere * 3)(a, b, c)
    assert L.orc_list_hit(arr, 2, O.v3(0, 0, 0), O.v3(0, 0, -1), 0.001, INF, C.byref(rec)) and rec.index == 0  # first wins ties
    assert L.orc_list_hit(arr, 3, O.v3(0, 0, 0), O.v3(0, 0, -1), 0.001, INF, C.byref(rec)) and rec.index == 2 and rec.t == 0.75
    arr2 = (O.Sphere * 3)(c, a, b)
    assert L.orc_list_hit(arr2, 3, O.v3(0, 0, 0), O.v3(0, 0, -1), 0.001, INF, C.byref(rec)) and rec.index == 0
    assert not L.orc_list_hit(arr, 0, O.v3(0, 0, 0), O.v3(0, 0, -1), 0.001, INF, C.byref(rec))  # empty list


# ------------------------------------------------------------ vector helpers --

def test_reflect_refract_schlick(L):
    assert O.vec_op("reflect", (1, -1, 0), (0, 1, 0)) == (1.0, 1.0, 0.0)
    out = O.D3()
    L.orc_v_refract(O.v3(0, -1, 0), O.v3(0, 1, 0), 1.0, out)        # normal incidence, eta 1: unchanged
    assert tuple(out) == (0.0, -1.0, 0.0)
    L.orc_v_refract(O.v3(0, -1, 0), O.v3(0, 1, 0), 1.0 / 1.5, out)  # normal incidence never bends
    assert tuple(out) == (0.0, -1.0, 0.0)
    s = math.sin(math.radians(30.0))
    L.orc_v_refract(O.v3(s, -math.cos(math.radians(30.0)), 0), O.v3(0, 1, 0), 1.0 / 1.5, out)
    assert abs(out[0] - s / 1.5) < 1e-15 and abs(math.hypot(out[0], out[1]) - 1.0) < 1e-15   # Snell
    assert abs(L.orc_reflectance(1.0, 1.5) - 0.04) < 1e-16          # r0 = ((1-1.5)/(1+1.5))^2
    assert L.orc_reflectance(0.0, 1.5) == 1.0
    assert L.orc_v_near_zero(O.v3(9e-9, -9e-9, 0)) == 1 and L.orc_v_near_zero(O.v3(1e-8, 0, 0)) == 0


def test_rejection_samplers_consume_in_order(L):
    # cube point (0.9,0.9,0.9) has |p|^2 > 1 -> rejected; next (0.5,0,0) accepted and normalised
    u = lambda x: (x + 1.0) / 2.0
    r = O.script_rng([u(0.9), u(0.9), u(0.9), u(0.5), u(0.0), u(0.0)])
    out = O.D3()
    L.orc_random_unit_vector(C.byref(r), out)
    assert tuple(out) == (1.0, 0.0, 0.0) and r.script_pos == 6
    r = O.script_rng([u(0.8), u(0.8), u(0.6), u(-0.7)])             # disk: |p|^2 = 1.28 rejected, 0.85 accepted
    L.orc_random_in_unit_disk(C.byref(r), out)
    assert r.script_pos == 4 and abs(out[0] - 0.6) < 1e-15 and abs(out[1] + 0.7) < 1e-15 and out[2] == 0.0
    r = O.script_rng([u(1.0), u(0.0), u(0.25), u(0.25)])            # |p|^2 == 1 is rejected (strict, vec3.rs:114)
    L.orc_random_in_unit_disk(C.byref(r), out)
    assert r.script_pos == 4 and tuple(out) == (0.25, 0.25, 0.0)
    r = O.script_rng([u(1.0 - 2 ** -23), u(0.0)])
    L.orc_random_in_unit_disk(C.byref(r), out)                       # |p|^2 < 1 strictly
    assert r.script_pos == 2
    # Philox layout: unit-vector attempt a reads uniforms 4a..4a+2
    r = O.philox_rng(9, 1, 2, 3)
    L.orc_random_unit_vector(C.byref(r), out)
    assert r.draws % 3 == 0
    assert abs(math.sqrt(sum(x * x for x in out)) - 1.0) < 1e-15


# ------------------------------------------------------------------ scatter --

def scatter(L, s, rd, rec, script):
    r = O.script_rng(script)
    att, so, sd = O.D3(), O.D3(), O.D3()
    ok = L.orc_scatter(C.byref(s), O.v3(*rd), C.byref(rec), C.byref(r), att, so, sd)
    return ok, tuple(att), tuple(so), tuple(sd), r.script_pos


def test_scatter_known_answers(L):
    u = lambda x: (x + 1.0) / 2.0
    lam = O.make_sphere((0, 0, -1), 0.5, O.MAT_LAMBERTIAN, (0.1, 0.2, 0.5))
    ok, rec = hit(L, lam, (0, 0, 0), (0, 0, -1))
    ok, att, so, sd, used = scatter(L, lam, (0, 0, -1), rec, [u(0.0), u(0.5), u(0.0)])
    assert ok and att == (0.1, 0.2, 0.5) and so == (0, 0, -0.5) and sd == (0.0, 1.0, 1.0) and used == 3
    ok, att, so, sd, used = scatter(L, lam, (0, 0, -1), rec, [u(0.0), u(0.0), u(-1.0)])  # n + (-n) is near zero -> n
    assert sd == (0.0, 0.0, 1.0)
    met = O.make_sphere((0, 0, -1), 0.5, O.MAT_METAL, (0.8, 0.6, 0.2), 0.0)
    ok, rec = hit(L, met, (0, 0, 0), (0, 0, -1))
    ok, att, so, sd, used = scatter(L, met, (0, 0, -1), rec, [u(0.5), u(0.0), u(0.0)])
    assert ok and sd == (0.0, 0.0, 1.0) and att == (0.8, 0.6, 0.2) and used == 3      # the unit vector is drawn even at fuzz 0
    fuzzy = O.make_sphere((0, 0, -1), 0.5, O.MAT_METAL, (0.8, 0.6, 0.2), 7.0)
    assert fuzzy.fuzz_or_ior == 1.0                                                   # clamp (material.rs:67)
    ok, att, so, sd, used = scatter(L, fuzzy, (0, 0, -1), rec, [u(0.0), u(0.0), u(-1.0)])
    assert not ok and sd == (0.0, 0.0, 0.0)                                           # reflected + (-n) -> dot = 0 -> absorbed
    gl = O.make_sphere((0, 0, -1), 0.5, O.MAT_DIELECTRIC, (0, 0, 0), 1.5)
    ok, rec = hit(L, gl, (0, 0, 0), (0, 0, -1))
    ok, att, so, sd, used = scatter(L, gl, (0, 0, -1), rec, [0.5])    # reflectance 0.04 < 0.5 -> refract straight on
    assert ok and att == (1.0, 1.0, 1.0) and sd == (0.0, 0.0, -1.0) and used == 1
    ok, att, so, sd, used = scatter(L, gl, (0, 0, -1), rec, [0.01])   # 0.04 > 0.01 -> reflect
    assert sd == (0.0, 0.0, 1.0) and used == 1
    # total internal reflection: leaving glass at a grazing angle draws NO uniform (short circuit, material.rs:114)
    rec_in = O.Hit()
    rec_in.p = O.v3(0, 0, -0.5)
    rec_in.normal = O.v3(0, 0, -1)
    rec_in.front_face = 0
    d = (math.sin(math.radians(60)), 0.0, math.cos(math.radians(60)))
    ok, att, so, sd, used = scatter(L, gl, d, rec_in, [0.99])
    assert ok and used == 0 and abs(sd[0] - d[0]) < 1e-15 and abs(sd[2] + d[2]) < 1e-15
    absorb = O.make_sphere((0, 0, -1), 0.5, O.MAT_DEFAULT)
    assert scatter(L, absorb, (0, 0, -1), rec, [])[0] == 0


# ------------------------------------------------------------ camera, colour --

def test_camera_defaults_and_initialize(L):
    cam = O.default_camera()                                          # camera.rs:147-163
    assert (cam.aspect_ratio, cam.image_width, cam.samples_per_pixel, cam.max_depth) == (1.0, 100, 10, 10)
    assert cam.vfov == 90.0 and tuple(cam.lookat) == (0, 0, -1) and tuple(cam.vup) == (0, 1, 0)
    assert cam.defocus_angle == 0.0 and cam.focus_dist == 10.0
    d = O.camera_initialize(O.camera_final())                         # main.rs:55-66
    assert d.image_height == 675
    assert d.viewport_height == 3.5265396141692995
    assert d.viewport_width == 6.2694037585231985
    assert d.defocus_radius == 0.052360356057001275
    assert d.pixel_samples_scale == 1.0 / 500
    for v in (d.u, d.v, d.w):
        assert abs(math.sqrt(sum(x * x for x in v)) - 1.0) < 1e-15
    tiny = O.default_camera()
    tiny.image_width, tiny.aspect_ratio = 1, 16.0
    assert O.camera_initialize(tiny).image_height == 1                # clamped to at least 1
    # pixel (0,0) with zero offsets looks at pixel00_loc
    cam = O.camera_three_spheres()
    d = O.camera_initialize(cam)
    r = O.script_rng([0.5, 0.5])
    ro, rd = O.D3(), O.D3()
    L.orc_get_ray(C.byref(cam), C.byref(d), 0, 0, C.byref(r), ro, rd)
    assert tuple(ro) == (0, 0, 0) and tuple(rd) == tuple(d.pixel00_loc) and r.script_pos == 2   # no aperture draws


def test_sky_and_colour(L):
    col = O.D3()
    empty = (O.Sphere * 1)()
    for dirn, want in (((0, 1, 0), (0.5, 0.7, 1.0)), ((0, -1, 0), (1.0, 1.0, 1.0)), ((0, 0, -3), (0.75, 0.85, 1.0))):
        r = O.script_rng([])
        L.orc_ray_color_recursive(empty, 0, O.v3(0, 0, 0), O.v3(*dirn), 5, C.byref(r), col, None)
        assert tuple(col) == want
    L.orc_ray_color_recursive(empty, 0, O.v3(0, 0, 0), O.v3(0, 1, 0), 0, C.byref(O.script_rng([])), col, None)
    assert tuple(col) == (0, 0, 0)                                    # depth <= 0 (camera.rs:265-267)
    rgb = (C.c_uint32 * 3)()
    L.orc_color_to_rgb(O.v3(1, 1, 1), rgb)
    assert list(rgb) == [255, 255, 255]
    L.orc_color_to_rgb(O.v3(0.25, 0.0, -1.0), rgb)
    assert list(rgb) == [127, 0, 0]
    L.orc_color_to_rgb(O.v3(float("nan"), 4.0, 1e-12), rgb)
    assert list(rgb) == [0, 255, 0]
    color = np.zeros(6, dtype=np.uint32)
    raw = np.zeros((6, 3))
    cnt = np.zeros(6, dtype=np.int32)
    P = C.POINTER
    packed = L.orc_write_color(1, 1, 3, O.v3(0.25, 0.0, 0.0), color.ctypes.data_as(P(C.c_uint32)),
                               raw.ctypes.data_as(P(C.c_double)), cnt.ctypes.data_as(P(C.c_int32)))
    assert packed == 0xFF7F0000 and color[4] == 0xFF7F0000 and cnt[4] == 1 and raw[4, 0] == 0.25
    L.orc_write_color(1, 1, 3, O.v3(0.75, 1.0, 0.0), color.ctypes.data_as(P(C.c_uint32)),
                      raw.ctypes.data_as(P(C.c_double)), cnt.ctypes.data_as(P(C.c_int32)))
    assert cnt[4] == 2 and raw[4, 0] == 1.0 and color[4] == (0xFF000000 | (int(255.999 * math.sqrt(0.5)) << 16) | (int(255.999 * math.sqrt(0.5)) << 8))


# ------------------------------------------------------------------- renders --

def test_render_orders_and_products_agree():
    sp = O.scene_three_spheres()
    cam = O.camera_three_spheres(width=48, spp=3, depth=10)
    a = O.render(cam, sp, seed=5)                       # sample-major, one thread (camera.rs:217-219)
    b = O.render(cam, sp, seed=5, threads=4)            # rows in parallel, pixel-major
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and (a[2] == b[2]).all() and a[3].segments == b[3].segments
    c = O.render(cam, sp, seed=5, iterative=1)          # ray_color_iteration's left-to-right product
    assert a[3].segments == c[3].segments
    np.testing.assert_allclose(a[1], c[1], rtol=1e-14)
    assert (a[0] == c[0]).all()
    # rows of an interleaved partition reproduce the full render
    color = np.zeros_like(a[0]); raw = np.zeros_like(a[1]); cnt = np.zeros_like(a[2])
    for r in range(3):
        O.render(cam, sp, seed=5, row_begin=r, row_step=3, color=color, raw=raw, count=cnt)
    assert (color == a[0]).all() and (raw == a[1]).all() and (cnt == 3).all()
    # accumulating semantics: 2 + 1 samples == 3 samples
    cam2 = O.camera_three_spheres(width=48, spp=2, depth=10)
    col, rw, cn, _ = O.render(cam2, sp, seed=5)
    cam1 = O.camera_three_spheres(width=48, spp=1, depth=10)
    O.render(cam1, sp, seed=5, sample_offset=2, color=col, raw=rw, count=cn)
    assert (col == a[0]).all() and (rw == a[1]).all()


def test_golden_fixtures_match_oracle():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "render_three_spheres_64x36.npz"))
    col, raw, cnt, st = O.render(O.camera_three_spheres(width=64, spp=4, depth=10), g["spheres"], seed=int(g["seed"]))
    assert (col == g["color"]).all() and (raw == g["raw"]).all() and st.segments == int(g["segments"])
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "render_final_96x54.npz"))
    assert (O.scene_final(1) == g["spheres"]).all()
    col, raw, cnt, st = O.render(O.camera_final(width=96, spp=8, depth=50), g["spheres"], seed=int(g["seed"]), threads=8)
    assert (col == g["color"]).all() and (raw == g["raw"]).all() and st.segments == int(g["segments"])
    txt = np.loadtxt(os.path.join(os.path.dirname(__file__), "golden", "scene_final_seed1.txt"))
    sp = g["spheres"]
    assert len(txt) == len(sp) == 485
    assert (txt[:, 0:3] == sp["center"]).all() and (txt[:, 3] == sp["radius"]).all() and (txt[:, 4] == sp["kind"]).all()
    assert (txt[:, 5:8] == sp["albedo"]).all() and (txt[:, 8] == sp["fuzz_or_ior"]).all()


def test_hit_scatter_vectors_match_oracle(L):
    rows = np.load(os.path.join(os.path.dirname(__file__), "golden", "hit_scatter_vectors.npy"))
    assert rows.shape[0] == 120
    for row in rows:
        kind, radius = int(row[0]), row[1]
        s = O.make_sphere(row[2:5], radius, kind, row[5:8], row[8])
        ro, rd = row[9:12], row[12:15]
        ok, rec = hit(L, s, ro, rd)
        assert ok == int(row[15])
        if not ok:
            continue
        assert rec.t == row[16] and tuple(rec.p) == tuple(row[17:20]) and tuple(rec.normal) == tuple(row[20:23])
        assert rec.front_face == int(row[23])
        u = list(row[24:36])
        sc, att, so, sd, used = scatter(L, s, rd, rec, u)
        assert sc == int(row[36]) and att == tuple(row[37:40]) and sd == tuple(row[40:43]) and used == int(row[43])


def test_scene_final_recipe():
    sp = O.scene_final(1)
    assert sp[0]["radius"] == 1000.0 and tuple(sp[0]["center"]) == (0.0, -1000.0, 0.0) and sp[0]["kind"] == O.MAT_LAMBERTIAN
    assert [tuple(s["center"]) for s in sp[-3:]] == [(0.0, 1.0, 0.0), (-4.0, 1.0, 0.0), (4.0, 1.0, 0.0)]
    assert [int(s["kind"]) for s in sp[-3:]] == [O.MAT_DIELECTRIC, O.MAT_LAMBERTIAN, O.MAT_METAL]
    small = sp[1:-3]
    assert (small["radius"] == 0.2).all() and (small["center"][:, 1] == 0.2).all()
    d = small["center"] - np.array([4.0, 0.2, 0.0])
    assert (np.sqrt((d * d).sum(1)) > 0.9).all()                      # main.rs:21
    assert 400 < len(small) <= 484
    metal = small[small["kind"] == O.MAT_METAL]
    assert (metal["albedo"] >= 0.5).all() and (metal["albedo"] < 1.0).all() and (metal["fuzz_or_ior"] < 0.5).all()
    assert (small[small["kind"] == O.MAT_DIELECTRIC]["fuzz_or_ior"] == 1.5).all()
    assert len(O.scene_final(2)) != 0 and not (O.scene_final(2)["center"][1] == sp["center"][1]).all()


def test_ray_color_iteration_steps_equal_the_whole_path(oracle):
    """Driving ray_color_iteration (camera.rs:118-144) to completion from the iterator's rays (camera.rs:103-114)
    reproduces the left-to-right product render: the sum of the finished entries' colours, in sample order, equals
    the raw buffer of the iterative render bit for bit, and the number of intersected entries equals its segments."""
    spheres = oracle.scene_final(1)
    cam = oracle.camera_final(width=24, spp=3, depth=12)
    H, W = oracle.image_height(cam), cam.image_width
    _, raw, _, st = oracle.render(cam, spheres, seed=5, iterative=1)
    px = np.array([(i, j, s) for s in range(3) for j in range(H) for i in range(W)], dtype=np.int32)
    e = oracle.camera_ray_entries(cam, px, seed=5)
    assert (e["depth"] == 12).all() and (e["stage"] == 1).all() and (e["attenuation"] == 1.0).all()
    assert (e["finished"] == 0).all()
    segments = 0
    for _ in range(13):
        live = int((e["finished"] == 0).sum())
        if not live:
            break
        before = e.copy()
        oracle.ray_color_iteration(e, spheres, W, seed=5)
        segments += int(((before["finished"] == 0) & (before["depth"] > 0)).sum())
        moved = (before["finished"] == 0) & (e["finished"] == 0)
        assert (e["depth"][moved] == before["depth"][moved] - 1).all()
        assert (e["stage"][moved] == before["stage"][moved] + 1).all()
        assert (e["hit_object_or_stop"][moved] == 1).all()
    assert (e["finished"] == 1).all()
    assert segments == st.segments
    total = np.zeros((H, W, 3))
    for k in range(len(e)):  # sample-major order of px: each pixel adds its samples in order, as colors.rs:38 does
        total[e["pixel_j"][k], e["pixel_i"][k]] += 0.0 + e["color"][k]
    assert (total == raw).all()
    # RayColorStatus: sky misses report hit_object_or_stop = false (camera.rs:143), everything else true
    sky = e["hit_object_or_stop"] == 0
    assert sky.any() and (e["color"][sky].sum(axis=1) > 0).all()
    # depth exhausted: one more call finishes the entry with colour attenuation * 0 (camera.rs:120-123)
    z = oracle.camera_ray_entries(cam, px[:1], seed=5)
    z["depth"] = 0
    oracle.ray_color_iteration(z, spheres, W, seed=5)
    assert z["finished"][0] == 1 and z["hit_object_or_stop"][0] == 1 and (z["color"][0] == 0).all()


def test_transparent_hands_the_ray_back(oracle):
    """Transparent::scatter (material.rs:22-31): attenuation (1, 1, 1), scattered = the incoming ray, origin included.
    A path that meets a Transparent sphere therefore meets it again at every depth it has left and ends black
    (camera.rs:265-267); ray_color_iteration shows the steps: same ray, one depth less, attenuation unchanged."""
    spheres = oracle.scene_three_spheres()
    spheres["kind"][1] = 4            # the centre sphere, straight ahead of the camera
    cam = oracle.camera_three_spheres(width=16, spp=1, depth=6)
    H, W = oracle.image_height(cam), cam.image_width
    px = np.array([(W // 2, H // 2, 0)], dtype=np.int32)
    e = oracle.camera_ray_entries(cam, px, seed=3)
    o0, d0 = e["origin"][0].copy(), e["direction"][0].copy()
    for step in range(6):
        oracle.ray_color_iteration(e, spheres, W, seed=3)
        assert e["finished"][0] == 0 and e["hit_object_or_stop"][0] == 1
        assert (e["origin"][0] == o0).all() and (e["direction"][0] == d0).all()
        assert e["depth"][0] == 5 - step and e["stage"][0] == 2 + step and (e["attenuation"][0] == 1.0).all()
    oracle.ray_color_iteration(e, spheres, W, seed=3)   # no depth left: black
    assert e["finished"][0] == 1 and (e["color"][0] == 0.0).all()
    # whole renders: the pixels of that sphere are black in both product orders, and each of their samples costs
    # max_depth segments
    solid = oracle.scene_three_spheres()
    for iterative in (0, 1):
        col, raw, _, st = oracle.render(cam, spheres, seed=3, iterative=iterative)
        assert (raw[H // 2, W // 2] == 0.0).all() and (col[H // 2, W // 2] & 0xFFFFFF) == 0
        _, raw2, _, st2 = oracle.render(cam, solid, seed=3, iterative=iterative)
        assert raw2[H // 2, W // 2].sum() > 0 and st.segments != st2.segments


def test_bounding_box_hit_as_written(L):
    """BoundingBox::hit (bounding_box.rs:50-76) is restated as written, quirks included; each value below follows
    from the cited lines by hand."""
    box = O.make_box(1.0, 2.0, -1.0, 1.0, -3.0, 3.0)
    rec = O.Hit()
    hit = lambda o, d, tmin=0.001, tmax=INF: L.orc_box_hit(C.byref(box), O.v3(*o), O.v3(*d), tmin, tmax, C.byref(rec))
    # straight at the box along x: axis 0 gives 1 - 0/1 = 1 and 2 - 0/1 = 2; axis 1 has direction 0 and is skipped (:54)
    assert hit((0, 0, 0), (1, 0, 0)) and rec.t == 1.0 and tuple(rec.p) == (1.0, 0.0, 0.0)
    # :55-56 compute `min - origin / direction`, not `(min - origin) / direction`: from x = 2 with direction 2 the
    # planes are "at" 1 - 2/2 = 0 and 2 - 2/2 = 1; 0 is not inside (0.001, inf), so t_max = 1 is taken (:68-72) and
    # the hit point lies at x = 4, outside the box
    assert hit((2, 0, 0), (2, 0, 0)) and rec.t == 1.0 and tuple(rec.p) == (4.0, 0.0, 0.0)
    # :53 loops over axes 0 and 1 only: z never matters -- a ray far outside the z interval still hits ...
    assert hit((0, 0, 100), (1, 0, 0)) and rec.t == 1.0 and tuple(rec.p) == (1.0, 0.0, 100.0)
    # ... and a ray along z, whose x and y directions are both 0, keeps t_min = -inf, t_max = +inf: neither is
    # inside the interval, so it misses even when it runs through the middle of the box
    assert not hit((1.5, 0, -10), (0, 0, 1))
    # both axes: max of the entries, min of the exits.  o = (0, -4), d = (1, 2): x planes 1, 2; y planes -1 + 2 = 1, 1 + 2 = 3
    assert hit((0, -4, 0), (1, 2, 0)) and rec.t == 1.0
    # y planes at -1 - (-12/2) = 5 and 7, x planes at 1 and 2: t_max = 2 < t_min = 5, a miss (:61)
    assert not hit((0, -12, 0), (1, 2, 0))
    # the interval is open (interval.rs:28-30): t_min == tmin falls through to t_max; an upper bound at or below t_min
    # leaves nothing (t_max >= t_min is no nearer)
    assert hit((0, 0, 0), (1, 0, 0), tmin=1.0) and rec.t == 2.0
    assert not hit((0, 0, 0), (1, 0, 0), tmax=1.0)
    assert hit((0, 0, 0), (1, 0, 0), tmax=1.5) and rec.t == 1.0
    # only t and p are written: normal and front_face keep what the record held (:63-66)
    rec.normal = O.v3(7.0, 8.0, 9.0)
    rec.front_face = 1
    assert hit((0, 0, 0), (1, 0, 0)) and tuple(rec.normal) == (7.0, 8.0, 9.0) and rec.front_face == 1
    # BoundingBox::default is three EMPTY intervals (+inf, -inf): t_min = -inf, t_max = +inf, nothing inside the interval
    empty = O.make_box(INF, -INF, INF, -INF, INF, -INF)
    assert not L.orc_box_hit(C.byref(empty), O.v3(0, 0, 0), O.v3(1, 1, 1), 0.001, INF, C.byref(rec))
    # a NaN direction is "not zero" (:54) and f64::max / f64::min drop the NaN operand: the bounds stay infinite
    assert not hit((0, 0, 0), (float("nan"), 0, 0))


def test_bounding_box_in_a_list_and_its_material(L, oracle):
    """A box takes part in HittableList::hit like any hittable (hittable_list.rs:65-70): smallest t, on equal t the
    entry inserted first.  Its material is Transparent (bounding_box.rs:65,71): the incoming ray comes back unchanged,
    meets the box again at every depth left, and the sample ends black."""
    box = O.make_box(1.0, 2.0, -1.0, 1.0, -3.0, 3.0)
    near = O.make_sphere((0.5, 0, 0), 0.25, O.MAT_LAMBERTIAN, (1, 0, 0))   # hit at t = 0.25
    same = O.make_sphere((1.5, 0, 0), 0.5, O.MAT_LAMBERTIAN, (0, 1, 0))    # hit at t = 1.0, the box's distance
    rec = O.Hit()
    o, d = O.v3(0, 0, 0), O.v3(1, 0, 0)
    assert L.orc_list_hit((O.Sphere * 2)(box, near), 2, o, d, 0.001, INF, C.byref(rec)) and rec.index == 1 and rec.t == 0.25
    assert L.orc_list_hit((O.Sphere * 2)(box, same), 2, o, d, 0.001, INF, C.byref(rec)) and rec.index == 0 and rec.t == 1.0
    assert L.orc_list_hit((O.Sphere * 2)(same, box), 2, o, d, 0.001, INF, C.byref(rec)) and rec.index == 0
    # the sphere's normal stays in the record when the box behind it in the list is hit closer (the reference's
    # temp_rec is shared by the whole scan, hittable_list.rs:61-69)
    far = O.make_sphere((5, 0, 0), 0.5, O.MAT_LAMBERTIAN, (0, 0, 1))
    assert L.orc_list_hit((O.Sphere * 2)(far, box), 2, o, d, 0.001, INF, C.byref(rec)) and rec.index == 1
    assert tuple(rec.normal) == (-1.0, 0.0, 0.0) and tuple(rec.p) == (1.0, 0.0, 0.0)
    # whole paths: a box that fills the view makes every sample black after max_depth segments
    spheres = oracle.scene_three_spheres()
    arr = np.concatenate([spheres, oracle.spheres_to_numpy((O.Sphere * 1)(O.make_box(-50, 50, -50, 50, -2, -1)), 1)])
    cam = oracle.camera_three_spheres(width=16, spp=2, depth=5)
    for iterative in (0, 1):
        col, raw, cnt, st = oracle.render(cam, arr, seed=4, iterative=iterative)
        _, raw0, _, st0 = oracle.render(cam, spheres, seed=4, iterative=iterative)
        assert raw0.sum() > 0
        # x planes "at" -50 - 0/dx and 50 - 0/dx (the camera is at the origin): every ray is inside from t = -50 to 50,
        # so t_max = 50 is inside the interval unless a sphere is closer: the sky pixels turn black, spheres stay
        sky = raw0[0, 0].sum() > 0 and raw[0, 0].sum() == 0
        assert sky and st.segments > st0.segments
