#!/usr/bin/env python3
"""The headline frame in passes through a render context (page-locked buffers), a few times over: the run to put under
`rocprofv3 --kernel-trace --memory-copy-trace` when the question is how the passes overlap (tools/trace_timeline.py).
    python tools/ctx_passes_trace.py [--passes 10] [--rounds 3]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--passes", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    import dray_amd as d
    world = d.scene_final(1)
    W = 1200
    cam = d.camera_final(width=W, spp=500, depth=50)
    ctx = d.RenderContext([0])
    b = ctx.alloc_buffers(W * W)
    cam.context = ctx
    cam.samples_per_pass = -(-500 // args.passes)
    for _ in range(args.rounds):
        t0 = time.perf_counter()
        cam.render(world, None, *b)
        print("%.3f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
