#!/usr/bin/env python3
"""Grid scan against hybrid scan on recipe scenes of growing size (which scan should DRAY_SCAN_AUTO pick?).

    python tools/crossover.py [--spp 100]
"""
import argparse
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, default=100)
    ap.add_argument("--halves", type=int, nargs="+", default=[2, 3, 4, 5, 6, 8, 11])
    args = ap.parse_args()
    import dray_amd as d
    from dray_amd import device as ddev
    for h in args.halves:
        spheres = d.scene_recipe(1, -h, h, 0.8, 0.15, 0.5)
        cam = d.camera_final(width=1200, spp=args.spp, depth=50)
        cam.mode = d.MODE_F32
        scene = ddev.DeviceScene(spheres, 0)
        color, raw, count = ddev.alloc_buffers(cam, 0)
        row = {"spheres": len(spheres)}
        for name, v in (("hybrid", d.SCAN_HYBRID), ("grid", d.SCAN_GRID)):
            ddev.render_device(cam, scene, color, raw, count, variant=v)
            sts = [ddev.render_device(cam, scene, color, raw, count, variant=v) for _ in range(3)]
            row[name + "_ms"] = round(statistics.median([s.kernel_ms for s in sts]), 3)
            row["segments"] = int(sts[0].segments)
            if v == d.SCAN_GRID:
                row["walk_steps"] = int(sts[0].walk_steps)
        print(json.dumps(row))
        scene.close()


if __name__ == "__main__":
    main()
