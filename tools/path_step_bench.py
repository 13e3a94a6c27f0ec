#!/usr/bin/env python3
"""Throughput of the resumable path step (dray_camera_rays_device / dray_path_step_device, the batch form of
CameraRayIterator::next and ray_color_iteration, camera.rs:103-144) with the entries resident in HBM: one entry per
pixel of a frame, stepped until every path has ended.  Prints, per scene and scan, the first step (every entry is
intersected) and the whole run as entries per second, timed with events on the stream the kernels run on.

    python tools/path_step_bench.py [--scenes final grid10k] [--width 1200] [--modes f64 f32]
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", nargs="+", default=["final", "grid10k"])
    ap.add_argument("--width", type=int, default=1200)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--modes", nargs="+", default=["f64", "f32"])
    ap.add_argument("--variants", type=int, nargs="+", default=[0, 1], help="0 grid walk, 1 whole list")
    args = ap.parse_args()
    import torch
    import dray_amd as d
    from dray_amd import _ffi
    from dray_amd import device as ddev
    L = _ffi.lib()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev)
    for name in args.scenes:
        spheres = {"final": d.scene_final, "grid10k": d.scene_grid_10k, "glass_metal": d.scene_glass_metal}[name](1)
        cam = d.camera_final(width=args.width, spp=1, depth=args.depth)
        if name == "grid10k":
            cam.lookfrom, cam.focus_dist = (13.0, 6.0, 3.0), 14.0
        cam.initialize()
        W, H = int(cam.image_width), int(cam.image_height)
        n = W * H
        order = np.random.default_rng(0).permutation(n)
        px = np.stack([order % W, order // W, np.zeros_like(order)], axis=1).astype(np.int32)
        d_px = torch.from_numpy(px).to(dev)
        scene = ddev.DeviceScene(spheres, 0)
        pod = cam.pod()
        for mode in args.modes:
            for variant in args.variants:
                if mode == "f32" and variant == 1:
                    continue
                opts = cam.opts(variant=variant, mode=d.MODE_F32 if mode == "f32" else d.MODE_F64)
                opts.device = 0
                d_entries = torch.zeros(n * _ffi.PATH_ENTRY_DTYPE.itemsize, dtype=torch.uint8, device=dev)
                best_first, best_all, steps, segs = 1e9, 1e9, 0, 0
                for rep in range(3):
                    _ffi.check(L.dray_camera_rays_device(C.addressof(pod), d_px.data_ptr(), n, d_entries.data_ptr(),
                                                         C.addressof(opts), C.c_void_p(stream.cuda_stream)))
                    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                    ev[0].record(stream)
                    _ffi.check(L.dray_path_step_device(scene.handle, W, d_entries.data_ptr(), n, C.addressof(opts),
                                                       C.c_void_p(stream.cuda_stream)))
                    ev[1].record(stream)
                    steps = 1
                    for _ in range(args.depth):  # every path has ended after max_depth + 1 steps
                        _ffi.check(L.dray_path_step_device(scene.handle, W, d_entries.data_ptr(), n, C.addressof(opts),
                                                           C.c_void_p(stream.cuda_stream)))
                        steps += 1
                    ev[2].record(stream)
                    torch.cuda.synchronize(dev)
                    best_first = min(best_first, ev[0].elapsed_time(ev[1]))
                    best_all = min(best_all, ev[0].elapsed_time(ev[2]))
                host = np.frombuffer(d_entries.cpu().numpy().tobytes(), dtype=_ffi.PATH_ENTRY_DTYPE)
                assert host["finished"].all()
                segs = int((args.depth - host["depth"]).sum() + (host["hit_object_or_stop"] == 0).sum())
                print(json.dumps({"scene": name, "spheres": len(spheres), "entries": n, "mode": mode,
                                  "scan": "grid" if variant == 0 else "whole list",
                                  "first_step_ms": round(best_first, 4), "first_step_entries_per_s": round(n / best_first * 1e3),
                                  "steps": steps, "all_steps_ms": round(best_all, 3), "segments": segs,
                                  "segments_per_s": round(segs / best_all * 1e3)}))
        scene.close()


if __name__ == "__main__":
    main()
