#!/bin/bash
# One rocprofv3 counter pass over tools/ab_bench.py (any scene); prints per-dispatch averages of the trace kernel.
# usage: tools/pmc_ab.sh <name> "<counters>" [ab_bench.py arguments...]
name=$1; ctrs=$2; shift 2
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/pmc_$name
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d "$out" -o c -- python3 $root/tools/ab_bench.py --rounds 1 "$@" > "$out/ab.json" 2> "$out/err.txt" || { tail -5 "$out/err.txt"; exit 1; }
python3 - "$out" <<'PY'
import csv, sys, json, collections
vals = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1] + "/c_counter_collection.csv")):
    if "render_f32" in r["Kernel_Name"]:
        vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(json.dumps({k: sum(v) / len(v) for k, v in vals.items()}))
print(open(sys.argv[1] + "/ab.json").read().strip().splitlines()[-1][:300])
PY
