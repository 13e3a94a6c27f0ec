#!/bin/bash
# Counter passes over one full-size frame of the headline config for one scan variant; one rocprofv3 run per pass
# (at most 8 SQ counters each), merged into <outdir>/counters_v<variant>.csv.  Stops at the first failing pass.
# usage: tools/pmc_final.sh <outdir> <variant>
out=$1; v=$2
root=$GRAFT_REPO_ROOT
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
passes=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
 "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
 "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F64"
)
k=0
for ctrs in "${passes[@]}"; do
  timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d "$out" -o v${v}_p$k -- python3 $root/bench.py --steps 1 --warmup 0 --variant $v --no-cpu-baseline > "$out/bench_v${v}_p$k.json" 2> "$out/err_v${v}_p$k.txt" || { tail -5 "$out/err_v${v}_p$k.txt"; exit 1; }
  k=$((k+1))
done
python3 - "$out" "$v" <<'PY'
# sums over the dispatches of this variant's kernel (one per pass, since every pass runs one step without warm-up)
import csv, sys, json, glob
out, v = sys.argv[1], sys.argv[2]
vals = {}
for f in sorted(glob.glob(f"{out}/v{v}_p*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if f"render_f32_kernel<{v}>" in r["Kernel_Name"]:  # bench.py also times the brute-force kernel
            vals[r["Counter_Name"]] = vals.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
b = json.loads(open(f"{out}/bench_v{v}_p0.json").read().strip().splitlines()[-1])
with open(f"{out}/counters_v{v}.csv", "w") as o:
    o.write(f"# kernel_ms under the first pass {b['roofline']['kernel_ms']:.2f}; segments per sample {b['segments_per_sample']:.4f}\n")
    o.write("counter,value\n")
    for k in sorted(vals): o.write(f"{k},{vals[k]:.6g}\n")
print(open(f"{out}/counters_v{v}.csv").read())
PY
