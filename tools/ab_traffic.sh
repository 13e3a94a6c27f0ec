#!/bin/bash
# A/B of the in-tree library against a baseline build (trace-kernel time at 1200x675x500), then the trace
# kernel's FETCH_SIZE / WRITE_SIZE (KiB per launch), each counter in a rocprofv3 pass of its own.
# usage: tools/ab_traffic.sh <tag> [baseline.so]
tag=$1; base=$2
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/traffic_$tag
mkdir -p "$out"
timeout -k 10 120 python3 $root/tools/ab_bench.py --spp 500 --rounds 3 --tag $tag || exit 1
if [ -n "$base" ]; then DRAY_HIP_LIB=$base timeout -k 10 120 python3 $root/tools/ab_bench.py --spp 500 --rounds 3 --tag base || exit 1; fi
cd /tmp && export TMPDIR=/tmp
for c in WRITE_SIZE FETCH_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d "$out" -o $c -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> "$out/err_$c.txt" || exit 1
done
python3 - "$out" <<'PY'
import csv, glob, json, sys
res = {}
for f in sorted(glob.glob(sys.argv[1] + "/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        # bench.py times the default (grid, <3>) kernel and then the brute-force (hybrid, <2>) one: keep them apart,
        # and take each kernel's first dispatch
        short = ("render_f32_kernel<3>" if "render_f32_kernel<3>" in name else "render_f32_kernel<2>" if "render_f32_kernel<2>" in name
                 else "render_f32_kernel<1>" if "render_f32" in name else "resolve_f32_kernel" if "resolve" in name else None)
        if short:
            res.setdefault(short, {}).setdefault(r["Counter_Name"] + "_KiB", float(r["Counter_Value"]))
for k, v in res.items():
    # FETCH_SIZE counts a wide coalesced read at half its bytes on gfx950: doubled here; WRITE_SIZE is taken as is
    v["bytes_per_launch"] = 2 * 1024 * v.get("FETCH_SIZE_KiB", 0.0) + 1024 * v.get("WRITE_SIZE_KiB", 0.0)
    print(k, v)
json.dump({"command": "bench.py --steps 1 --warmup 0 --no-cpu-baseline (1200x675, 500 spp, depth 50, seed 1)",
           "passes": "one rocprofv3 --pmc pass per counter", "kernels": res}, open(sys.argv[1] + "/traffic.json", "w"), indent=1)
PY
