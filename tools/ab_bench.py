#!/usr/bin/env python3
"""A/B timing of f32 scan variants or f64 organisations (and, via DRAY_HIP_LIB, of two builds) in ONE process on one GPU:
interleaved rounds, median and minimum trace-kernel time per arm.

    python tools/ab_bench.py --variants 0 1 2 --rounds 5 [--spp 100] [--width 1200]
"""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", type=int, nargs="+", default=[0])
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--spp", type=int, default=100)
    ap.add_argument("--width", type=int, default=1200)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--ppp", type=int, nargs="+", default=[0], help="pixels per pool values to try")
    ap.add_argument("--scene", default="final", choices=["final", "three", "grid10k", "glass_metal"])
    ap.add_argument("--tag", default="")
    ap.add_argument("--mode", default="f32", choices=["f32", "f64"],
                    help="f64: the variants are the f64 render's organisations (0/3 by wave, 2 by pixel, 1 whole list)")
    args = ap.parse_args()
    import dray_amd as d
    from dray_amd import device as ddev
    spheres = {"final": d.scene_final, "grid10k": d.scene_grid_10k, "glass_metal": d.scene_glass_metal}.get(
        args.scene, lambda seed: d.scene_three_spheres())(1)
    cam = (d.camera_three_spheres if args.scene == "three" else d.camera_final)(width=args.width, spp=args.spp,
                                                                                 depth=args.depth)
    cam.mode = d.MODE_F64 if args.mode == "f64" else d.MODE_F32
    scene = ddev.DeviceScene(spheres, 0)
    color, raw, count = ddev.alloc_buffers(cam, 0)
    arms = [(v, s) for v in args.variants for s in args.ppp]
    times = {a: [] for a in arms}
    segs = {}
    for a in arms:  # warm-up
        ddev.render_device(cam, scene, color, raw, count, variant=a[0], pixels_per_pool=a[1])
    for _ in range(args.rounds):
        for a in arms:
            st = ddev.render_device(cam, scene, color, raw, count, variant=a[0], pixels_per_pool=a[1])
            times[a].append(st.kernel_ms)
            segs[a] = st.segments
            steps = st.walk_steps
    for a in arms:
        med, mn = statistics.median(times[a]), min(times[a])
        tf = 18.0 * len(spheres) * segs[a] / (med * 1e-3) / 1e12
        print(json.dumps({"tag": args.tag, "lib": os.environ.get("DRAY_HIP_LIB", "in-tree"), "variant": a[0],
                          "pixels_per_pool": a[1], "median_ms": round(med, 3), "min_ms": round(mn, 3),
                          "mrays_s": round(segs[a] / med / 1e3, 1), "algo_tflops": round(tf, 2),
                          "frac_of_157.3": round(tf / 157.3, 4), "walk_steps": int(steps)}))
    scene.close()


if __name__ == "__main__":
    main()
