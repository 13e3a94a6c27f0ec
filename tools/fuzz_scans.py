#!/usr/bin/env python3
"""Random scenes through the three sphere-scan formulations of the f32 kernel: the images, sums and segment counts of
the grid and hybrid scans must equal the scalar scan's (exact test on every sphere) bit for bit.  With --mode f64:
through the three organisations of the f64 render, likewise.

    python tools/fuzz_scans.py [--scenes 100] [--seed 0]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def random_scene(rng, d):
    n = int(rng.choice([64, 100, 300, 1000, 3000]))
    layout = rng.choice(["cloud", "flat", "line", "clustered", "shell"])
    scale = float(10.0 ** rng.uniform(-1, 2))
    offset = rng.uniform(-1, 1, 3) * scale * float(rng.choice([0, 0, 1, 30]))
    sp = np.zeros(n, dtype=d._ffi.SPHERE_DTYPE)
    if layout == "cloud":
        c = rng.uniform(-1, 1, (n, 3))
    elif layout == "flat":
        c = rng.uniform(-1, 1, (n, 3)) * (1, 0.02, 1)
    elif layout == "line":
        c = rng.uniform(-1, 1, (n, 3)) * (1, 0.01, 0.01)
    elif layout == "clustered":
        c = rng.normal(0, 0.05, (n, 3)) + rng.uniform(-1, 1, (8, 3))[rng.integers(0, 8, n)]
    else:
        v = rng.normal(size=(n, 3))
        c = v / np.linalg.norm(v, axis=1, keepdims=True)
    radius = 10.0 ** rng.uniform(-2.5, -1.0, n) if rng.random() < 0.5 else rng.uniform(0.005, 0.08, n)
    k = int(rng.integers(0, 12))                       # outsized spheres: below, at and above the loose-list limit
    if k:
        radius[rng.integers(0, n, k)] = rng.uniform(0.2, 0.9, k)
    if rng.random() < 0.3:
        radius[int(rng.integers(0, n))] = 50.0 / scale  # one routed to the f64 test
    if rng.random() < 0.3:
        c[int(rng.integers(0, n))] = c[int(rng.integers(0, n))]  # concentric pair
    sp["center"] = c * scale + offset
    sp["radius"] = radius * scale
    sp["kind"] = rng.integers(0, 4, n)
    sp["albedo"] = rng.uniform(0.1, 1, (n, 3))
    sp["fuzz_or_ior"] = np.where(sp["kind"] == 3, 1.5, rng.uniform(0, 1, n))
    cam = d.Camera.new()
    cam.aspect_ratio, cam.image_width = 1.5, 48
    cam.samples_per_pixel, cam.max_depth = int(rng.choice([2, 4])), int(rng.choice([4, 12]))
    where = rng.choice(["outside", "inside", "far"])
    r = {"outside": 2.5, "inside": 0.5, "far": 40.0}[where]
    v = rng.normal(size=3)
    cam.lookfrom = tuple(offset + scale * r * v / np.linalg.norm(v))
    cam.lookat = tuple(offset + scale * rng.uniform(-0.3, 0.3, 3))
    cam.vfov = float(rng.choice([20, 60, 100])) if where != "far" else 4.0
    cam.defocus_angle = float(rng.choice([0.0, 0.6]))
    cam.focus_dist = scale * r
    cam.mode = d.MODE_F32
    cam.seed = int(rng.integers(1, 1 << 30))
    return sp, cam, "%s n=%d scale=%.2g offset=%.2g outsized=%d %s" % (layout, n, scale, np.abs(offset).max(), k, where)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=100)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--mode", default="f32", choices=["f32", "f64"],
                    help="f64: the f64 render's three organisations (whole list, one thread per pixel, stream of samples), "
                         "both product orders in turn")
    args = ap.parse_args()
    import dray_amd as d
    from dray_amd import device as ddev
    rng = np.random.default_rng(args.seed)
    bad = 0
    for k in range(args.scenes):
        sp, cam, what = random_scene(rng, d)
        if args.mode == "f64":
            cam.mode, cam.flags = d.MODE_F64, (d.FLAG_ITERATIVE_PRODUCT if k % 2 else 0)
        scene = ddev.DeviceScene(sp, 0)
        res = {}
        # "memory": the grid scan (f64: the stream kernel) with the tables read from memory even where a copy fits in LDS
        for variant in (d.SCAN_SCALAR, d.SCAN_HYBRID, d.SCAN_GRID, "memory"):
            color, raw, count = ddev.alloc_buffers(cam, 0)
            if variant == "memory":
                os.environ["DRAY_GRID_LDS"] = "0"
            st = ddev.render_device(cam, scene, color, raw, count, variant=d.SCAN_GRID if variant == "memory" else variant)
            os.environ.pop("DRAY_GRID_LDS", None)
            res[variant] = (ddev.color_to_numpy(color).copy(), raw.cpu().numpy().copy(), st.segments)
        scene.close()
        ok = all((res[v][0] == res[d.SCAN_SCALAR][0]).all() and (res[v][1] == res[d.SCAN_SCALAR][1]).all()
                 and res[v][2] == res[d.SCAN_SCALAR][2] for v in res)
        info = d.grid_inspect(sp)[0]
        if not ok:
            bad += 1
        if not ok or k % 10 == 0:
            print("%s scene %d: %s grid %s loose %d wide %d segments %d" % ("ok " if ok else "BAD", k, what, list(info.n),
                                                                              info.n_loose, info.n_wide, res[d.SCAN_SCALAR][2]), flush=True)
    print("%d scenes, %d mismatches" % (args.scenes, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
