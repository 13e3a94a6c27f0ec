#!/bin/bash
# Full-size records of BASELINE.json's GPU configurations: for each, the bench line, a kernel trace of the same
# command and the counter passes; for configuration 5 also the run with the wave-level compaction switched off.
# Results land under gpurun_out/configs/ and gpurun_out/counters/.  Stops at the first failing step.
# usage: tools/config_records.sh [configs...]   (default: 2 4 5)
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/configs
mkdir -p "$out"
cfgs=${@:-2 4 5}
for c in $cfgs; do
  cd $root
  timeout -k 10 400 python3 bench.py --config $c > "$out/bench_c$c.json" 2> "$out/bench_c$c.err" || { tail -5 "$out/bench_c$c.err"; exit 1; }
  echo "config $c: $(python3 -c "import json,sys; d=json.loads(open('$out/bench_c$c.json').read().strip().splitlines()[-1]); print(round(d['value']), 'Mrays/s', round(d['ms_per_step'],2), 'ms')")"
  if [ "$c" = "5" ]; then
    timeout -k 10 400 python3 bench.py --config 5 --no-refill --no-extra-legs --no-cpu-baseline > "$out/bench_c5_norefill.json" 2> "$out/bench_c5_norefill.err" || { tail -5 "$out/bench_c5_norefill.err"; exit 1; }
    echo "config 5, compaction off: $(python3 -c "import json; d=json.loads(open('$out/bench_c5_norefill.json').read().strip().splitlines()[-1]); print(round(d['value']), 'Mrays/s', round(d['ms_per_step'],2), 'ms')")"
  fi
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt_c$c" -o kt -- python3 $root/bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > "$out/bench_c${c}_under_rocprof.json" 2> "$out/kt_c$c.err" || { tail -5 "$out/kt_c$c.err"; exit 1; }
  cd $root && tools/kernel_counters.sh $c > "$out/counters_c$c.txt" 2>&1 || { tail -5 "$out/counters_c$c.txt"; exit 1; }
done
