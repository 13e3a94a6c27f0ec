#!/usr/bin/env python3
"""Where a wave of the f64 stream kernel spends its time: shader clocks per section of the main loop, summed over
waves, from a diagnostic build with -DDRAY_F64_CLOCKS
(tools/build_variant_lib.sh clocks render_f64 "-DDRAY_F64_CLOCKS" -> tools/ab_libs/clocks.so).

    DRAY_HIP_LIB=tools/ab_libs/clocks.so python tools/f64_clocks.py [--config 3] [--spp N]
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SECTIONS = ["walk", "hit", "end_sample", "retire", "refill", "philox_block0", "coop_draw", "scatter", "camera_ray", "new_ray"]
COUNTS = {16: "turns", 17: "lanes_with_sample", 18: "walk_steps", 19: "walking_lanes_at_steps", 20: "lanes_shading", 21: "settle_turns", 22: "first_slots",
          23: "settle_turns_with_second_slot", 24: "second_slots"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--spp", type=int, default=0)
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
    cam.mode = d.MODE_F64
    L = d._ffi.lib()
    L.dray_debug_f64_clocks.argtypes = [C.c_void_p]
    scene = ddev.DeviceScene(spheres, 0)
    color, raw, count = ddev.alloc_buffers(cam, 0)
    ddev.render_device(cam, scene, color, raw, count)
    buf = (C.c_ulonglong * 32)()
    L.dray_debug_f64_clocks(buf)          # clear what the warm-up counted
    st = ddev.render_device(cam, scene, color, raw, count, sample_offset=cam.samples_per_pixel)
    assert L.dray_debug_f64_clocks(buf) == 0
    total = sum(int(buf[k]) for k in range(len(SECTIONS))) or 1
    out = {"kernel_ms": st.kernel_ms, "segments": int(st.segments), "config": args.config, "spp": cam.samples_per_pixel,
           "share": {n: round(int(buf[k]) / total, 4) for k, n in enumerate(SECTIONS)}}
    out.update({n: int(buf[k]) for k, n in COUNTS.items()})
    turns = max(out["turns"], 1)
    out["per_turn"] = {"clocks": total / turns, "lanes_with_sample": out["lanes_with_sample"] / turns,
                       "lanes_shading": out["lanes_shading"] / turns, "walk_steps": out["walk_steps"] / turns,
                       "walking_lanes_per_step": out["walking_lanes_at_steps"] / max(out["walk_steps"], 1)}
    print(json.dumps(out))
    scene.close()


if __name__ == "__main__":
    main()
