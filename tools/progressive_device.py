#!/usr/bin/env python3
"""The headline frame on device buffers in 1 ... 50 passes (a progressive render: the sums stay in HBM, every pass
leaves a displayable image), one call at a time against enqueued back to back (dray_render_device_async).

    python tools/progressive_device.py
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dray_amd as d
from dray_amd import device as ddev
spheres = d.scene_final(1)
scene = ddev.DeviceScene(spheres, 0)
cam = d.camera_final()
cam.mode = d.MODE_F32
color, raw, count = ddev.alloc_buffers(cam, 0)
def run(passes, pipelined):
    cam.samples_per_pixel = 500 // passes
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ts = []
    for p in range(passes):
        ts.append(ddev.render_device(cam, scene, color, raw, count, sample_offset=p * cam.samples_per_pixel,
                                     want_stats="later" if pipelined else True))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
for passes in (1, 2, 5, 10, 20, 50):
    for pipelined in (False, True):
        run(passes, pipelined)
        ms = sorted(run(passes, pipelined) for _ in range(5))[2]
        print(json.dumps({"passes": passes, "pipelined": pipelined, "ms": round(ms, 3)}))
