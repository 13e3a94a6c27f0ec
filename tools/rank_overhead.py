#!/usr/bin/env python3
"""Per-step host overhead of one rank of a G-rank job, measured on one GPU without a process group:
wall time per DistributedRenderer.step() against the trace-kernel time, for rank 0 of G.

    python tools/rank_overhead.py [--ranks 1 2 4 8]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--steps", type=int, default=12)
    args = ap.parse_args()
    import torch
    import dray_amd as d
    from dray_amd import device as ddev
    from dray_amd import partition
    spheres = d.scene_final(1)
    scene = ddev.DeviceScene(spheres, 0)
    for G in args.ranks:
        cam = d.camera_final()
        cam.mode = d.MODE_F32
        r = partition.DistributedRenderer(cam, partition.gpu_render_rows(scene), partition.gpu_alloc(0), rank=0,
                                          world_size=G)
        for pipelined in (False, True):
            r.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            k = 0.0
            segs = 0
            tickets = []
            def collect(st):
                nonlocal k, segs
                st = st.stats() if pipelined else st
                k += st.kernel_ms
                segs += st.segments

            for _ in range(args.steps):
                _, st = r.step(stats_later=pipelined)   # pipelined: steps enqueued back to back, looked at a few steps later
                tickets.append(st)
                if len(tickets) > 6:                    # (a scene answers for its last 16 calls)
                    collect(tickets.pop(0))
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) * 1e3 / args.steps
            for st in tickets:
                collect(st)
            print(json.dumps({"ranks": G, "pipelined": pipelined, "wall_ms_per_step": round(wall, 3),
                              "kernel_ms": round(k / args.steps, 3), "overhead_ms": round(wall - k / args.steps, 3),
                              "mrays_s_if_all_ranks_alike": round(G * segs / args.steps / wall / 1e3, 1)}))
    scene.close()


if __name__ == "__main__":
    main()
