#!/bin/bash
# One rocprofv3 counter pass over any python command of this repository; prints every kernel's counters, averaged
# over its dispatches.  usage: tools/pmc_cmd.sh <name> "<counters>" <script.py> [arguments...]
name=$1; ctrs=$2; script=$3; shift 3
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/pmc_$name
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc $ctrs --output-format csv -d "$out" -o c -- python3 $root/$script "$@" > "$out/stdout.txt" 2> "$out/err.txt" || { tail -5 "$out/err.txt"; exit 1; }
python3 - "$out" <<'PY'
import csv, sys, json, glob, re
vals, disp = {}, {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", ""))
        vals.setdefault(k, {})
        vals[k][r["Counter_Name"]] = vals[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        disp.setdefault(k, set()).add(r["Dispatch_Id"])
for k in vals:
    n = len(disp[k])
    print(json.dumps({"kernel": k, "dispatches": n, **{c: v / n for c, v in sorted(vals[k].items())}}))
PY
tail -3 "$out/stdout.txt"
