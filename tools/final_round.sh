#!/bin/bash
# Everything the round's records need, on the sources as they are: counters for the four GPU configurations, kernel
# traces, bench lines, the compaction on/off pair, the one-call timings, the launch timeline, one rank's share of a
# G-rank job, the converged-image check with independent seeds, the path step's throughput and the two-rank
# rehearsal of the headline frame at full size.  usage: tools/final_round.sh
root=$GRAFT_REPO_ROOT
cd $root
tools/profile_round.sh final || exit 1
out=$root/gpurun_out/final
python3 tools/trace_timeline.py $(ls $out/kt/*kernel_trace.csv | head -1) 40 > $out/timeline.txt || exit 1
tools/config_records.sh 2 4 5 || exit 1
timeout -k 10 200 python3 tools/context_timing.py > $out/context_timing.json || exit 1
timeout -k 10 200 python3 tools/rank_overhead.py > $out/rank_overhead.jsonl || exit 1
timeout -k 10 200 python3 tools/independent_seed_psnr.py > $out/independent_seed_psnr.json || exit 1
timeout -k 10 200 python3 tools/path_step_bench.py > $out/path_step_bench.jsonl || exit 1
timeout -k 10 200 python3 tools/fullsize_parity.py > $out/fullsize_parity.txt || exit 1
DRAY_BENCH_REHEARSAL=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --verify-image > $out/rehearsal_2ranks_fullsize.json 2> $out/rehearsal_err.txt || { tail -5 $out/rehearsal_err.txt; exit 1; }
if [ -f tools/ab_libs/sections.so ]; then
  DRAY_HIP_LIB=tools/ab_libs/sections.so timeout -k 10 200 python3 tools/section_counts.py --config 3 > $out/sections_c3.json || exit 1
fi
echo done
