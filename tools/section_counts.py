#!/usr/bin/env python3
"""How often each section of the f32 trace kernel's main loop runs on one frame (wave level), from a diagnostic
build of the library with -DDRAY_SECTION_COUNTERS (tools/build_sections_lib.sh -> tools/ab_libs/sections.so).

    DRAY_HIP_LIB=tools/ab_libs/sections.so python tools/section_counts.py [--config 3] [--spp N]
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = ["iterations", "iterations_with_scan_setup", "walk_steps", "walking_lanes_at_steps", "lanes_shading",
         "iterations_with_shading", "iterations_with_refill", "coop_draw_rounds", "iterations_with_block0", "camera_rays"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--no-refill", action="store_true")
    args = ap.parse_args()
    import bench
    import dray_amd as d
    from dray_amd import device as ddev
    cfg = bench.CONFIGS[args.config]
    spheres = {"final": d.scene_final, "grid10k": d.scene_grid_10k, "glass_metal": d.scene_glass_metal,
               "three": lambda seed: d.scene_three_spheres()}[cfg["scene"]](1)
    cam = (d.camera_three_spheres if cfg["scene"] == "three" else d.camera_final)(
        width=cfg["width"], spp=args.spp or cfg["spp"], depth=cfg["depth"])
    if cfg["scene"] == "grid10k":
        cam.lookfrom, cam.focus_dist = (13.0, 6.0, 3.0), 14.0
    cam.flags = d.FLAG_NO_REFILL if args.no_refill else 0
    L = d._ffi.lib()
    L.dray_debug_section_counts.argtypes = [C.c_void_p]
    scene = ddev.DeviceScene(spheres, 0)
    color, raw, count = ddev.alloc_buffers(cam, 0)
    ddev.render_device(cam, scene, color, raw, count)
    buf = (C.c_ulonglong * 16)()
    L.dray_debug_section_counts(buf)          # clear what the warm-up counted
    st = ddev.render_device(cam, scene, color, raw, count, sample_offset=cam.samples_per_pixel)
    assert L.dray_debug_section_counts(buf) == 0
    out = {n: int(buf[k]) for k, n in enumerate(NAMES)}
    out.update(segments=int(st.segments), samples=int(st.samples), kernel_ms=st.kernel_ms, config=args.config,
               spp=cam.samples_per_pixel, refill=not args.no_refill)
    it = max(out["iterations"], 1)
    out["per_iteration"] = {"segments": out["segments"] / it, "lanes_shading": out["lanes_shading"] / it,
                            "walk_steps": out["walk_steps"] / it, "coop_rounds": out["coop_draw_rounds"] / it,
                            "walking_lanes_per_step": out["walking_lanes_at_steps"] / max(out["walk_steps"], 1)}
    print(json.dumps(out))
    scene.close()


if __name__ == "__main__":
    main()
