#!/usr/bin/env python3
"""Merge the rocprofv3 counter CSVs of tools/kernel_counters.sh into one JSON per configuration, and (with --into)
fold such files into profiles/counters.json, the file bench.py reads its counter-derived roofline fields from.

    python tools/merge_counters.py <pass dir> <config> <out.json>
    python tools/merge_counters.py --into profiles/counters.json gpurun_out/counters/config3.json [...]
"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def short_name(name):
    m = re.search(r"render_f32_kernel<(?:\(int\))?(\d), (?:\(bool\))?(?:false|0)>", name)   # the kernels without boxes
    if m:
        return "render_f32_kernel<%s>" % m.group(1)
    if "render_f64_stream_kernel" in name:
        return "render_f64_stream_kernel"
    if "render_ref_kernel<double>" in name:
        return "render_ref_kernel<double>"
    if "render_ref_kernel<float>" in name:
        return "render_ref_kernel<float>"
    if "resolve_f32_kernel" in name:
        return "resolve_f32_kernel"
    return None


def merge(pass_dir, config, out_path):
    from bench import source_hash
    kernels = {}
    for f in sorted(glob.glob(os.path.join(pass_dir, "**", "*counter_collection.csv"), recursive=True)):
        rows = [r for r in csv.DictReader(open(f)) if short_name(r["Kernel_Name"])]
        first = {}   # every kernel's first dispatch of the pass (bench.py launches some kernels more than once)
        for r in rows:
            k, disp = short_name(r["Kernel_Name"]), int(r["Dispatch_Id"])
            first[k] = min(first.get(k, disp), disp)
        sums = {}
        for r in rows:
            k = short_name(r["Kernel_Name"])
            if int(r["Dispatch_Id"]) == first[k]:   # a counter may come as several rows (one per instance): add them up
                sums[(k, r["Counter_Name"])] = sums.get((k, r["Counter_Name"]), 0.0) + float(r["Counter_Value"])
        for (k, name), v in sums.items():
            kernels.setdefault(k, {})[name] = v
    bench = None
    for f in sorted(glob.glob(os.path.join(pass_dir, "bench_p*.json"))):
        lines = [l for l in open(f).read().splitlines() if l.startswith("{")]
        if lines:
            bench = json.loads(lines[-1])
            break
    if bench:
        timed = bench["roofline"]["kernel"].split(" ")[0]
        # (the bench names the grid kernel from the committed counters: before the first pass over a scene whose tables
        # are staged in LDS it says <3> where <4> ran)
        if timed == "render_f32_kernel<3>" and timed not in kernels and "render_f32_kernel<4>" in kernels:
            timed = "render_f32_kernel<4>"
        if timed in kernels:
            kernels[timed]["kernel_ms_under_counters"] = bench["roofline"]["kernel_ms"]
        if "roofline_bruteforce" in bench and "render_f32_kernel<2>" in kernels and timed != "render_f32_kernel<2>":
            kernels["render_f32_kernel<2>"]["kernel_ms_under_counters"] = bench["roofline_bruteforce"]["kernel_ms"]
        if "f64_leg" in bench and "render_f64_stream_kernel" in kernels:
            kernels["render_f64_stream_kernel"]["kernel_ms_under_counters"] = bench["f64_leg"]["kernel_ms"]
    doc = {"config": int(config), "source_hash": source_hash(),
           "command": "bench.py --config %s --steps 1 --warmup 0 --no-cpu-baseline, one rocprofv3 --pmc pass per counter group; "
                      "values are per launch (each kernel's first dispatch of the pass)" % config,
           "workload": bench["config"]["workload"] if bench else None, "kernels": kernels}
    json.dump(doc, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps({k: {n: v[n] for n in ("SQ_INSTS_VALU", "kernel_ms_under_counters") if n in v} for k, v in kernels.items()}))


def fold(target, files):
    try:
        doc = json.load(open(target))
    except (OSError, ValueError):
        doc = {"file": "profiles/counters.json", "configs": {}}
    for f in files:
        one = json.load(open(f))
        doc["configs"][str(one["config"])] = {k: one[k] for k in ("source_hash", "command", "workload", "kernels")}
    json.dump(doc, open(target, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    if sys.argv[1] == "--into":
        fold(sys.argv[2], sys.argv[3:])
    else:
        merge(sys.argv[1], sys.argv[2], sys.argv[3])
