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
import csv, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "render_f32" in r["Kernel_Name"] or "resolve" in r["Kernel_Name"]:
            print(r["Kernel_Name"].split("(")[1][:30] if "anonymous" in r["Kernel_Name"] else r["Kernel_Name"][:40], r["Kernel_Name"][28:60], r["Counter_Name"], "KiB", r["Counter_Value"])
PY
