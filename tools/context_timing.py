#!/usr/bin/env python3
"""Wall time of the one-call route (host buffers in, host buffers out) on the headline frame: dray_render with its
cache cold and warm, a render context with staged and with registered buffers, and a progressive render in passes
against one pass.  Prints one JSON line.

    python tools/context_timing.py [--width 1200] [--spp 500] [--passes 10]
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1200)
    ap.add_argument("--spp", type=int, default=500)
    ap.add_argument("--passes", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=5)
    args = ap.parse_args()
    import dray_amd as d
    world = d.scene_final(1)
    W = args.width

    def buffers():
        return np.zeros(W * W, np.uint32), np.zeros((W * W, 3), np.float64), np.zeros(W * W, np.int32)

    def timed(cam, bufs, rounds=args.rounds):
        ts, kernel = [], []
        for _ in range(rounds):
            t0 = time.perf_counter()
            cam.render(world, None, *bufs)
            ts.append((time.perf_counter() - t0) * 1e3)
            kernel.append(cam.last_stats.kernel_ms)
        return {"median_ms": round(statistics.median(ts), 3), "min_ms": round(min(ts), 3), "kernel_ms": round(statistics.median(kernel), 3)}

    out = {"frame": "%dx%d, %d spp" % (W, W * 9 // 16, args.spp)}
    cam = d.camera_final(width=W, spp=args.spp, depth=50)
    bufs = buffers()
    d._ffi.lib().dray_release_cache()
    t0 = time.perf_counter()
    cam.render(world, None, *bufs)
    out["dray_render_first_call_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
    out["dray_render_warm"] = timed(cam, bufs)
    for name, pin in (("context_staged", False), ("context_pinned", True)):
        ctx = d.RenderContext([0])
        b = ctx.alloc_buffers(W * W) if pin else buffers()   # page-locked memory of the context, or plain numpy
        cam.context = ctx
        cam.render(world, None, *b)
        out[name] = timed(cam, b)
        cam.samples_per_pass = -(-args.spp // args.passes)
        out[name + "_%d_passes" % args.passes] = timed(cam, b)
        cam.samples_per_pass = 0
        cam.context = None
        ctx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
