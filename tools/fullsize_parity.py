#!/usr/bin/env python3
"""The headline frame (final scene 1200x675, 500 spp, depth 50) through the f64 reference-order kernel and through the
f32 throughput kernel, same Philox streams: per-channel 8-bit differences, segment counts, PSNR of the linear means.
The f64 kernel is the one the tests hold bit-equal to the CPU oracle on smaller frames.

    python tools/fullsize_parity.py [--spp 500] [--width 1200]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, default=500)
    ap.add_argument("--width", type=int, default=1200)
    ap.add_argument("--depth", type=int, default=50)
    args = ap.parse_args()
    import torch
    import dray_amd as d
    from dray_amd import device as ddev
    scene = ddev.DeviceScene(d.scene_final(1), 0)
    out = {}
    for name, mode in (("f64", d.MODE_F64), ("f32", d.MODE_F32)):
        cam = d.camera_final(width=args.width, spp=args.spp, depth=args.depth)
        cam.mode, cam.seed = mode, 1
        color, raw, count = ddev.alloc_buffers(cam, 0)
        t0 = time.perf_counter()
        st = ddev.render_device(cam, scene, color, raw, count)
        torch.cuda.synchronize()
        print("%s: %.2f s, %d segments" % (name, time.perf_counter() - t0, st.segments), flush=True)
        H, W = cam.image_height, cam.image_width
        c = ddev.color_to_numpy(color).reshape(-1)[:W * H]
        out[name] = (np.stack([(c >> 16) & 255, (c >> 8) & 255, c & 255], axis=-1).astype(np.int32),
                     raw.cpu().numpy().reshape(-1, 3)[:W * H] / args.spp, st.segments)
    a, b = out["f64"], out["f32"]
    diff = np.abs(a[0] - b[0])
    worst = diff.max(axis=-1)
    mse = float(np.mean((a[1] - b[1]) ** 2))
    print(json.dumps({
        "frame": "%dx%d, %d spp, depth %d" % (args.width, len(worst) // args.width, args.spp, args.depth),
        "pixels_identical": float((worst == 0).mean()), "pixels_within_1_lsb": float((worst <= 1).mean()),
        "max_lsb": int(worst.max()), "mean_abs_lsb": float(diff.mean()),
        "segments_f64": int(a[2]), "segments_f32": int(b[2]), "segments_ratio": b[2] / a[2],
        "psnr_linear_db": 10 * np.log10(1.0 / mse) if mse > 0 else float("inf")}))
    scene.close()


if __name__ == "__main__":
    main()
