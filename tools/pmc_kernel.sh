#!/bin/bash
# One rocprofv3 counter pass over a short bench run; prints the trace kernel's counters.
# usage: tools/pmc_kernel.sh <name> "<counters>" [bench.py arguments...]
name=$1; ctrs=$2; shift 2
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/pmc_$name
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d "$out" -o c -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > "$out/bench.json" 2> "$out/err.txt" || { tail -5 "$out/err.txt"; exit 1; }
python3 - "$out" <<'PY'
import csv, sys, json
vals = {}
for r in csv.DictReader(open(sys.argv[1] + "/c_counter_collection.csv")):
    if "render_f32" in r["Kernel_Name"]:
        vals[r["Counter_Name"]] = vals.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print(json.dumps(vals))
b = json.loads(open(sys.argv[1] + "/bench.json").read().strip().splitlines()[-1])
print(json.dumps({"kernel_ms": b["roofline"]["kernel_ms"], "segments_per_sample": b["segments_per_sample"], "value": b["value"]}))
PY
