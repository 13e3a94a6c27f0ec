#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (whole job) on the RTIOW final scene, 1200x675, 500 spp, depth 50.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {2,3,4,5}]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" is one full render of the frame: every rank traces its interleaved scanlines with the
f32 throughput kernel (scene, camera and frame buffers resident in HBM), the resolve kernel packs
the 8-bit image, and for N > 1 the slabs are all-gathered over RCCL.  A ray is one path segment
(one closest-hit query over the whole sphere list).  Rank 0 prints one JSON line.

The roofline block describes the kernel that is timed.  The default scan walks a uniform grid, so the
work it executes is not the reference's 18 x spheres FLOP per segment.  `roofline.achieved` / `frac` are
the FP32 arithmetic the kernel executes -- the FMA (2 FLOP), add, multiply and transcendental instruction
counters x 64 lanes x the active-lane share, from a committed counter pass over this very command
(profiles/counters.json) -- over the launch's duration, which is measured live with HIP events on the
launch stream, against the FP32 vector peak.  Most of a path tracer's instructions are compares, selects,
moves and integer work, so beside it stands the share of the vector unit's issue slots the kernel fills
(`valu_issue_frac`: executed vector instructions x 128 FLOP-equivalents per wave64 slot).  When the
committed counters were taken on other kernel sources than the ones being run, every counter-derived
field is null.  The 18 x spheres x segments figure of the reference's linear scan is reported for the
brute-force kernel (`roofline_bruteforce`, which executes every test) and, for the grid kernel, under
`reference_equivalent` only.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FLOP_PER_TEST = 18.0          # reject-path flops of one ray-sphere test (sphere.rs:29-36, |d|^2 hoisted)
PEAK_FP32_VALU = 157.3e12     # MI355X vector FP32 peak, FLOP/s: 1024 SIMDs x 32 lanes x 2 FLOP x 2.4 GHz
PEAK_FP64_VALU = 78.6e12      # vector FP64 peak
FLOP_PER_ISSUE_SLOT = 128.0   # one wave64 vector instruction = 64 lanes x 2 FLOP (an FMA) at the peak above
KERNEL_SOURCES = ("render_f32.hip", "render_ref.hip", "render_f64.hip", "ref_device.h", "dray_device.h", "dray_internal.h",
                  "dray_api.hip")

# BASELINE.json's configurations that run on a GPU: scene, camera, frame
CONFIGS = {
    2: dict(scene="three", width=1920, spp=100, depth=50,
            name="RTIOW 3-sphere scene (ground + Lambertian/metal/dielectric)"),
    3: dict(scene="final", width=1200, spp=500, depth=50, name="RTIOW final scene"),
    4: dict(scene="grid10k", width=3840, spp=256, depth=50, name="synthetic 10k-sphere scene (final-scene recipe over 100 x 100 cells)"),
    5: dict(scene="glass_metal", width=1200, spp=1000, depth=100,
            name="dielectric/metal-heavy scene (final-scene layout, half glass, half low-fuzz metal)"),
}


def source_hash():
    """Identifies the kernels a counter pass was taken on: sha256 over the library's kernel and launch sources."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "distributed-raytracer_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def committed_counters(config, kernel):
    """Counter values per launch of `kernel` under this command's configuration, from profiles/counters.json
    (written by tools/kernel_counters.sh: one rocprofv3 --pmc pass per counter group).  Counters cannot be read
    from inside this process.  Returns (values or None, whether they were taken on the sources as they are now)."""
    try:
        with open(os.path.join(ROOT, "profiles", "counters.json")) as f:
            doc = json.load(f)
        entry = doc["configs"][str(config)]
        return entry["kernels"][kernel], entry.get("source_hash") == source_hash(), doc.get("file", "profiles/counters.json")
    except (OSError, KeyError, ValueError):
        return None, False, None


def cpu_baseline(spheres, cam_of, width, depth, seed):
    """The CPU oracle (a port of the reference's single-threaded loop) timed on a bounded sample of the frame."""
    import _oracle as O
    cam = cam_of(O, width, 1, depth)
    t0 = time.perf_counter()
    O.render(cam, spheres, seed=seed, threads=1)                      # sizing pass: one sample per pixel
    t1 = time.perf_counter() - t0
    spp = max(1, min(16, int(round(12.0 / max(t1, 1e-3)))))           # aim at ~12 s of single-thread work
    cam.samples_per_pixel = spp
    t0 = time.perf_counter()
    _, _, _, st = O.render(cam, spheres, seed=seed, threads=1)
    dt = time.perf_counter() - t0
    one = {"value": st.segments / dt / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port",
           "sample": "same scene and camera, %dx%d at %d spp (%d samples, %d segments, %.1f s), 1 thread: the "
                     "reference's render loop is single-threaded" % (width, O.image_height(cam), spp, st.samples,
                                                                    st.segments, dt),
           "samples_per_s": st.samples / dt}
    cores = min(16, os.cpu_count() or 1)
    t0 = time.perf_counter()
    _, _, _, st = O.render(cam, spheres, seed=seed, threads=cores)
    dt = time.perf_counter() - t0
    many = {"value": st.segments / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "same sample, rows in parallel (OpenMP), %.1f s" % dt}
    return one, many


def executed_roofline(cnt, current, cfile, kernel_name, k_s, peak, f64=False):
    """Roofline block of a kernel from its committed counters and its live duration `k_s` (seconds per launch)."""
    out = {"bound": "valu", "achieved": None, "peak": peak / 1e12, "unit": "TFLOP/s", "frac": None, "traffic": None,
           "kernel": kernel_name, "kernel_ms": k_s * 1e3, "counters_file": cfile, "counters_current": bool(current)}
    if not cnt or k_s <= 0:
        out["note"] = "no counter pass for this configuration is committed: only the duration is measured"
        return out
    insts = cnt.get("SQ_INSTS_VALU")
    lanes = None
    if cnt.get("SQ_THREAD_CYCLES_VALU") and cnt.get("SQ_ACTIVE_INST_VALU"):
        lanes = cnt["SQ_THREAD_CYCLES_VALU"] / (64.0 * cnt["SQ_ACTIVE_INST_VALU"])
    if not current:
        # counters of other kernel sources say nothing about this run: only the duration stands
        out["note"] = ("the committed counter pass was taken on other kernel sources than these: the counter-derived fields "
                       "are left empty")
        return out
    if f64:
        # executed f64 arithmetic: an f64 FMA is 2 FLOP per lane; masked lanes are taken out with the lane share
        flop = 64.0 * (2.0 * cnt.get("SQ_INSTS_VALU_FMA_F64", 0.0) + cnt.get("SQ_INSTS_VALU_ADD_F64", 0.0) +
                       cnt.get("SQ_INSTS_VALU_MUL_F64", 0.0)) * (lanes if lanes else 1.0)
        out["achieved"] = flop / k_s / 1e12
        out["work"] = ("executed f64 FLOP per launch = 64 x (2 x SQ_INSTS_VALU_FMA_F64 + ADD_F64 + MUL_F64) x active-lane "
                       "share, over the live duration")
    else:
        fl = 64.0 * (2.0 * cnt.get("SQ_INSTS_VALU_FMA_F32", 0.0) + cnt.get("SQ_INSTS_VALU_ADD_F32", 0.0) +
                     cnt.get("SQ_INSTS_VALU_MUL_F32", 0.0) + cnt.get("SQ_INSTS_VALU_TRANS_F32", 0.0))
        if fl > 0:
            out["achieved"] = fl * (lanes if lanes else 1.0) / k_s / 1e12
            out["work"] = ("executed f32 FLOP per launch = 64 x (2 x SQ_INSTS_VALU_FMA_F32 + ADD_F32 + MUL_F32 + TRANS_F32) x "
                           "active-lane share, over the live duration")
        if insts:
            # the share of the vector unit's issue slots the kernel fills: what bounds it (compares, selects, moves and
            # integer work fill most of them)
            out["valu_insts_per_launch"] = insts
            out["valu_issue_tflops_equiv"] = insts * FLOP_PER_ISSUE_SLOT / k_s / 1e12
            out["valu_issue_frac"] = out["valu_issue_tflops_equiv"] / (peak / 1e12)
            if lanes:
                out["valu_issue_frac_active_lanes"] = out["valu_issue_frac"] * lanes
    if out["achieved"] is not None:
        out["frac"] = out["achieved"] / (peak / 1e12)
    if lanes:
        out["active_lane_share"] = lanes
    if cnt.get("GRBM_GUI_ACTIVE") and cnt.get("kernel_ms_under_counters"):
        out["clock_ghz_under_counters"] = cnt["GRBM_GUI_ACTIVE"] / 8.0 / (cnt["kernel_ms_under_counters"] * 1e-3) / 1e9
    if "FETCH_SIZE" in cnt and "WRITE_SIZE" in cnt:
        # KiB per launch; FETCH_SIZE counts a wide coalesced read at half its bytes on gfx950
        out["traffic"] = 2 * 1024.0 * cnt["FETCH_SIZE"] + 1024.0 * cnt["WRITE_SIZE"]
        out["traffic_unit"] = "HBM bytes per launch: FETCH_SIZE x 2 + WRITE_SIZE, one rocprofv3 --pmc pass each"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS), help="BASELINE.json configuration (3 = headline)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--depth", type=int, default=0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--mode", default="f32", choices=["f32", "f64", "f32plain"])
    ap.add_argument("--pixels-per-pool", type=int, default=0)
    ap.add_argument("--variant", type=int, default=0, help="f32 sphere-scan formulation: 0 auto, 1 scalar, 2 hybrid, 3 grid (identical images)")
    ap.add_argument("--no-refill", action="store_true", help="a lane whose path ends idles until its whole wave is done (compaction off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the brute-force and f64 legs")
    ap.add_argument("--verify-image", action="store_true", help="rank 0 also renders the whole frame alone and compares it with the gathered image")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    width, spp, depth = args.width or cfg["width"], args.spp or cfg["spp"], args.depth or cfg["depth"]
    full_size = (width, spp, depth, args.seed) == (cfg["width"], cfg["spp"], cfg["depth"], 1)

    import torch
    import torch.distributed as dist
    import dray_amd as d
    from dray_amd import device as ddev
    from dray_amd import partition

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    # Rehearsal on a box with fewer GPUs than ranks (never what the driver runs): DRAY_BENCH_REHEARSAL=1 puts every
    # rank on GPU 0 and swaps RCCL, which refuses two ranks on one device, for gloo; the gather is then staged
    # through host memory (partition.gather_image) and the timings mean nothing.
    rehearsal = os.environ.get("DRAY_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    make_scene = {"final": d.scene_final, "grid10k": d.scene_grid_10k, "glass_metal": d.scene_glass_metal,
                  "three": lambda seed: d.scene_three_spheres()}[cfg["scene"]]
    spheres = make_scene(args.seed)
    # The scene before the process group: a process has four hardware queues per GPU, HIP deals its streams out over
    # them as they are made, and the scene's two trace streams should each get one of their own (made after one or
    # three other streams they came to share one, and a step of a rank's eighth took 3.62 instead of 3.46 ms).
    scene = ddev.DeviceScene(spheres, local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    make_cam = d.camera_three_spheres if cfg["scene"] == "three" else d.camera_final
    cam = make_cam(width=width, spp=spp, depth=depth)
    if cfg["scene"] == "grid10k":  # look across the field of spheres rather than at its middle
        cam.lookfrom, cam.focus_dist = (13.0, 6.0, 3.0), 14.0
    cam.mode = {"f32": d.MODE_F32, "f64": d.MODE_F64, "f32plain": d.MODE_F32_PLAIN}[args.mode]
    cam.seed = args.seed
    flags = d.FLAG_NO_REFILL if args.no_refill else 0
    cam.flags = flags
    renderer = partition.DistributedRenderer(
        cam, partition.gpu_render_rows(scene, pixels_per_pool=args.pixels_per_pool, variant=args.variant),
        partition.gpu_alloc(local_rank), rank=rank, world_size=world)

    # ---- the legs beside the headline figure, before the timed steps (a GPU that has just run them flat out for
    # seconds clocks lower): the timed kernel alone, the brute-force scan, the f64 kernel ----
    n_spheres = len(spheres)
    scan = {0: "grid" if n_spheres - 1 >= 40 else "hybrid", 1: "scalar", 2: "hybrid", 3: "grid"}[args.variant]
    applies = world == 1 and full_size and args.pixels_per_pool == 0 and not args.no_refill
    extras = {}
    extra = world == 1 and not args.no_extra_legs and args.mode == "f32"
    if extra:
        # one pass of the timed kernel into fresh buffers: the image the other legs are compared with
        c32, r32, n32 = ddev.alloc_buffers(cam, local_rank)
        ddev.render_device(cam, scene, c32, r32, n32, pixels_per_pool=args.pixels_per_pool, variant=args.variant)   # warm-up
        c32.zero_(); r32.zero_(); n32.zero_()
        alone = ddev.render_device(cam, scene, c32, r32, n32, pixels_per_pool=args.pixels_per_pool, variant=args.variant)
        extras["alone"] = {"kernel_ms": alone.kernel_ms, "resolve_ms": alone.resolve_ms,
                           "note": "one launch of the timed kernel with nothing else on the GPU (in the timed steps a launch "
                                   "begins while the one before it runs out, which its own duration then includes)"}
    if extra and scan == "grid" and n_spheres <= 2048:
        # the same frame through the scan that performs all 18 x spheres x segments flops (identical image)
        cb, rb, nb = ddev.alloc_buffers(cam, local_rank)
        ddev.render_device(cam, scene, cb, rb, nb, variant=d.SCAN_HYBRID)   # warm-up
        cb.zero_(); rb.zero_(); nb.zero_()
        bst = ddev.render_device(cam, scene, cb, rb, nb, variant=d.SCAN_HYBRID)
        b_s = bst.kernel_ms / 1e3
        b_ach = FLOP_PER_TEST * n_spheres * bst.segments / b_s if b_s > 0 else 0.0
        extras["roofline_bruteforce"] = {
            "bound": "valu", "achieved": b_ach / 1e12, "peak": PEAK_FP32_VALU / 1e12, "unit": "TFLOP/s",
            "frac": b_ach / PEAK_FP32_VALU, "kernel": "render_f32_kernel<2> (hybrid scan: every test executed)",
            "kernel_ms": bst.kernel_ms, "mrays_s": bst.segments / b_s / 1e6 if b_s > 0 else 0.0,
            "work": "18 FLOP x spheres x segments of one launch, over its HIP-event duration",
            "image_equals_default_scan": bool((cb == c32).all().item() and (rb == r32).all().item())}
        del cb, rb, nb
    if extra:
        # the same frame at the reference's precision (vec3.rs:8-10): the f64 kernel (reference operation order, the
        # frame run as a stream of samples), one launch
        cam64 = make_cam(width=width, spp=spp, depth=depth)
        cam64.lookfrom, cam64.focus_dist, cam64.seed, cam64.mode = cam.lookfrom, cam.focus_dist, args.seed, d.MODE_F64
        c64, r64, n64 = ddev.alloc_buffers(cam64, local_rank)
        st64 = ddev.render_device(cam64, scene, c64, r64, n64)
        k64 = st64.kernel_ms / 1e3
        cnt64, cur64, cf64 = committed_counters(args.config, "render_f64_stream_kernel") if applies else (None, False, None)
        leg = executed_roofline(cnt64, cur64, cf64, "render_f64_stream_kernel<false>", k64, PEAK_FP64_VALU, f64=True)
        leg.update({"dtype": "f64", "mrays_s": st64.segments / k64 / 1e6 if k64 > 0 else 0.0, "segments": int(st64.segments),
                    "launches": 1})
        shifts = torch.tensor([16, 8, 0], device=c64.device)
        d8 = ((c64.view(-1, 1) >> shifts) & 255) - ((c32.view(-1, 1) >> shifts) & 255)
        worst = d8.abs().max(dim=1).values
        leg["f32_image_vs_this"] = {"within_1_lsb": float((worst <= 1).float().mean().item()), "max_lsb": int(worst.max().item())}
        extras["f64_leg"] = leg
        del c64, r64, n64, c32, r32, n32

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        renderer.step()
    sync()
    t0 = time.perf_counter()
    segs = samples = 0
    kernel_ms = resolve_ms = 0.0
    image = None
    # The steps are enqueued back to back (a step's statistics are collected a few steps later, when it has long ended), so
    # that a step's trace kernel can begin while the previous step's is running out and its gather is on the way.
    pending = []

    def collect(ticket):
        nonlocal segs, samples, kernel_ms, resolve_ms
        st = ticket.stats() if hasattr(ticket, "stats") else ticket
        segs += st.segments
        samples += st.samples
        kernel_ms += st.kernel_ms
        resolve_ms += st.resolve_ms

    for _ in range(args.steps):
        image, ticket = renderer.step(time_gather=True, stats_later=True)
        pending.append(ticket)
        if len(pending) > 6:
            collect(pending.pop(0))
    sync()
    elapsed = time.perf_counter() - t0
    for ticket in pending:
        collect(ticket)
    gather_ms = renderer.gather_ms() / max(args.steps, 1)
    my_kernel_ms = kernel_ms / max(args.steps, 1)

    stat_dev = "cpu" if rehearsal else "cuda"
    tot = torch.tensor([float(segs), float(samples), elapsed], dtype=torch.float64, device=stat_dev)
    per_rank = [(my_kernel_ms, gather_ms)]
    if world > 1:
        sums = tot.clone()
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        total_segs, total_samples, elapsed = sums[0].item(), sums[1].item(), mx[2].item()
        mine = torch.tensor([my_kernel_ms, gather_ms], dtype=torch.float64, device=stat_dev)
        everyone = torch.empty(world * 2, dtype=torch.float64, device=stat_dev)   # concatenated: rank r at [2r, 2r+2)
        dist.all_gather_into_tensor(everyone, mine)
        per_rank = [tuple(r) for r in everyone.view(world, 2).cpu().tolist()]
    else:
        total_segs, total_samples = float(segs), float(samples)

    if rank == 0:
        k_s = my_kernel_ms / 1e3                                 # this rank's average trace-kernel duration
        scan_id = {"scalar": 1, "hybrid": 2, "grid": 3}[scan]
        kernel = "render_f32_kernel<%d>" % scan_id if args.mode == "f32" else (
            "render_f64_stream_kernel" if args.mode == "f64" else "render_ref_kernel<float>")
        flops_per_launch = FLOP_PER_TEST * n_spheres * (segs / max(args.steps, 1))
        ref_equiv = flops_per_launch / k_s if k_s > 0 else 0.0
        # counters apply to the configuration they were taken on: one GPU, this config at full size, default options
        cnt, current, cfile = committed_counters(args.config, kernel) if applies else (None, False, None)
        if applies and args.mode == "f32" and scan == "grid":
            # a scene whose grid tables fit in LDS runs the kernel that stages them there, named <4> in a kernel trace
            cnt4, current4, cfile4 = committed_counters(args.config, "render_f32_kernel<4>")
            if cnt4:
                kernel, cnt, current, cfile = "render_f32_kernel<4>", cnt4, current4, cfile4
        headline = args.config == 3
        out = {
            "metric": "Mrays/sec (whole node) on RTIOW final scene 1200x675 500spp" if headline
                      else "Mrays/sec (whole node) on BASELINE.json configuration %d" % args.config,
            "value": total_segs / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": {"f32": "f32", "f64": "f64", "f32plain": "f32"}[args.mode],
            "data": "synthetic",
            "config": {"workload": "%s (%d spheres, Philox seed %d) %dx%d, %d spp, depth %d; "
                                   "interleaved scanlines over %d GPU(s), whole scene replicated, one all-gather "
                                   "of the packed image per step" % (cfg["name"], n_spheres, args.seed, width,
                                                                    cam.image_height, spp, depth, world),
                       "baseline_config": args.config, "spheres": n_spheres, "width": width, "height": cam.image_height,
                       "spp": spp, "max_depth": depth, "mode": args.mode, "variant": args.variant, "scan": scan,
                       "refill": not args.no_refill, "partition": "rows j %% %d" % world},
            "samples_per_s": total_samples / elapsed,
            "segments_per_sample": total_segs / max(total_samples, 1.0),
            "source_hash": source_hash(),
            "per_rank": {"kernel_ms": [r[0] for r in per_rank], "kernel_ms_min": min(r[0] for r in per_rank),
                         "kernel_ms_max": max(r[0] for r in per_rank), "gather_ms": [r[1] for r in per_rank],
                         "gather_ms_max": max(r[1] for r in per_rank)},
        }
        if args.mode == "f32" and scan != "grid" and n_spheres >= 64:
            # a brute-force scan executes every test of the reference's linear scan: 18 x spheres FLOP per segment
            # (on a list of a few spheres the tests are a sliver of the work: there the executed instructions count)
            out["roofline"] = {"bound": "valu", "achieved": ref_equiv / 1e12, "peak": PEAK_FP32_VALU / 1e12,
                               "unit": "TFLOP/s", "frac": ref_equiv / PEAK_FP32_VALU, "traffic": None, "kernel": kernel,
                               "kernel_ms": k_s * 1e3,
                               "work": "18 FLOP x spheres x segments of one launch (every test is executed), over the live duration"}
            if cnt and current and "FETCH_SIZE" in cnt and "WRITE_SIZE" in cnt:
                out["roofline"]["traffic"] = 2 * 1024.0 * cnt["FETCH_SIZE"] + 1024.0 * cnt["WRITE_SIZE"]
        else:
            out["roofline"] = executed_roofline(cnt, current, cfile, kernel, k_s,
                                                PEAK_FP64_VALU if args.mode == "f64" else PEAK_FP32_VALU,
                                                f64=args.mode == "f64")
            out["reference_equivalent"] = {
                "tflops": ref_equiv / 1e12, "x_peak": ref_equiv / PEAK_FP32_VALU,
                "note": "18 FLOP x spheres x segments, the work of the reference's linear scan (hittable_list.rs:65-70), over "
                        "this kernel's duration.  The kernel does not execute it (the grid walk finds the same closest hit "
                        "from a few tests), so this is a speed-up figure, not a utilisation"}
        if "alone" in extras:
            out["roofline"]["kernel_ms_alone"] = extras["alone"]["kernel_ms"]
            out["roofline"]["resolve_ms"] = extras["alone"]["resolve_ms"]
        out.update(extras)
        if args.verify_image:
            one = partition.DistributedRenderer(
                cam, partition.gpu_render_rows(scene, pixels_per_pool=args.pixels_per_pool, variant=args.variant),
                partition.gpu_alloc(local_rank), rank=0, world_size=1)
            for _ in range(args.warmup + args.steps):   # the gathered image has accumulated this many passes
                one.step(gather=False)
            torch.cuda.synchronize()
            out["image_equals_single_rank"] = bool((one.color[:one.H].to(image.device) == image).all().item())
        if world == 1 and not args.no_cpu_baseline:
            import _oracle as O
            cam_of = (lambda O, w, s, dp: O.camera_three_spheres(width=w, spp=s, depth=dp)) if cfg["scene"] == "three" \
                else (lambda O, w, s, dp: O.camera_final(width=w, spp=s, depth=dp))
            if cfg["scene"] == "grid10k":
                def cam_of(O, w, s, dp):
                    c = O.camera_final(width=w, spp=s, depth=dp)
                    c.lookfrom, c.focus_dist = O.v3(13.0, 6.0, 3.0), 14.0
                    return c
            cw = width if n_spheres < 2000 else max(64, width // 8)   # a bounded sample: ~12 s of one core
            one, many = cpu_baseline(spheres, cam_of, cw, depth, args.seed)
            out["cpu_baseline"] = one
            out["cpu_baseline_all_cores"] = many
        print(json.dumps(out))
    scene.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
