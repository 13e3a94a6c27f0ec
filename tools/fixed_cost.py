#!/usr/bin/env python3
"""Trace-kernel time of rank 0's share of a G-rank job against samples per pixel: the intercept of the line is what
a launch costs whatever its size (ramp-up, the drain of the deepest paths)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, nargs="+", default=[1, 8])
    ap.add_argument("--spp", type=int, nargs="+", default=[125, 250, 500, 1000])
    ap.add_argument("--depth", type=int, default=50)
    args = ap.parse_args()
    import dray_amd as d
    from dray_amd import device as ddev
    from dray_amd import partition
    scene = ddev.DeviceScene(d.scene_final(1), 0)
    for G in args.ranks:
        pts = []
        for spp in args.spp:
            cam = d.camera_final(spp=spp, depth=args.depth)
            cam.mode = d.MODE_F32
            r = partition.DistributedRenderer(cam, partition.gpu_render_rows(scene), partition.gpu_alloc(0), rank=0, world_size=G)
            r.step()
            ks = sorted(r.step()[1].kernel_ms for _ in range(5))
            pts.append((spp, ks[2]))
        (x0, y0), (x1, y1) = pts[0], pts[-1]
        slope = (y1 - y0) / (x1 - x0)
        print(json.dumps({"ranks": G, "depth": args.depth, "kernel_ms": dict(pts), "ms_per_100spp": round(100 * slope, 4),
                          "intercept_ms": round(y0 - slope * x0, 4)}))
    scene.close()


if __name__ == "__main__":
    main()
