#!/bin/bash
# Counter passes over ONE step of a bench.py configuration at full size, one rocprofv3 run per pass (at most 8 SQ
# counters, FETCH_SIZE and WRITE_SIZE in passes of their own), merged by tools/merge_counters.py into
# gpurun_out/counters/config<N>.json.  bench.py launches the timed kernel, the brute-force kernel and the f64
# kernel in that one step, so each pass sees all three.  Stops at the first failing pass.
# usage: tools/kernel_counters.sh <config> [extra bench.py arguments...]
cfg=$1; shift
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/counters/c$cfg
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$out/available.txt" 2>/dev/null
have() { # keeps the counters this rocprofv3 knows
  local keep=""
  for c in "$@"; do if grep -qw "$c" "$out/available.txt"; then keep="$keep $c"; fi; done
  echo $keep
}
passes=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
 "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
 "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F64"
 "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_FLAT SQ_INSTS_VMEM"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
k=0
for group in "${passes[@]}"; do
  ctrs=$(have $group)
  if [ -n "$ctrs" ]; then
    timeout -s KILL 150 rocprofv3 --pmc $ctrs --output-format csv -d "$out" -o p$k -- python3 $root/bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline "$@" > "$out/bench_p$k.json" 2> "$out/err_p$k.txt" || { tail -5 "$out/err_p$k.txt"; exit 1; }
    echo "pass $k ok: $ctrs"
  fi
  k=$((k+1))
done
python3 $root/tools/merge_counters.py "$out" $cfg "$root/gpurun_out/counters/config$cfg.json"
