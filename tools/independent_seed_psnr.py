#!/usr/bin/env python3
"""Converged-image check of the headline frame at full size with INDEPENDENT seeds: PSNR of the linear means (running
sum / samples) between the f32 throughput kernel and the f64 kernel, between two f64 renders (the Monte-Carlo noise
floor at this sample count) and between two f32 renders; and the mean image level of each mode.

    python tools/independent_seed_psnr.py [--spp 500] [--width 1200]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, default=500)
    ap.add_argument("--width", type=int, default=1200)
    ap.add_argument("--depth", type=int, default=50)
    args = ap.parse_args()
    import dray_amd as d
    from dray_amd import device as ddev
    scene = ddev.DeviceScene(d.scene_final(1), 0)
    means = {}
    for mode_name, mode in (("f64", d.MODE_F64), ("f32", d.MODE_F32)):
        for seed in (101, 202):
            cam = d.camera_final(width=args.width, spp=args.spp, depth=args.depth)
            cam.mode = mode
            color, raw, count = ddev.alloc_buffers(cam, 0)
            ddev.render_device(cam, scene, color, raw, count, seed=seed)
            H, W = cam.image_height, cam.image_width
            means[(mode_name, seed)] = raw.cpu().numpy().reshape(-1, 3)[:W * H] / args.spp

    def psnr(a, b):
        mse = float(np.mean((means[a] - means[b]) ** 2))
        return 10 * np.log10(1.0 / mse) if mse > 0 else float("inf")

    print(json.dumps({
        "frame": "final scene %dx%d, %d spp, depth %d, seeds 101 and 202" % (args.width, len(means[("f64", 101)]) // args.width,
                                                                            args.spp, args.depth),
        "psnr_db_f32_vs_f64": [psnr(("f32", 101), ("f64", 202)), psnr(("f32", 202), ("f64", 101))],
        "psnr_db_f64_vs_f64": psnr(("f64", 101), ("f64", 202)),
        "psnr_db_f32_vs_f32": psnr(("f32", 101), ("f32", 202)),
        "mean_level_f64": [float(means[("f64", s)].mean()) for s in (101, 202)],
        "mean_level_f32": [float(means[("f32", s)].mean()) for s in (101, 202)]}))
    scene.close()


if __name__ == "__main__":
    main()
