#!/bin/bash
# Copies what tools/final_round.sh left under gpurun_out/ into profiles/<round>/ under the names the design notes cite,
# and folds the counters into profiles/counters.json.  Run here (not on the GPU box) after a final_round call.
# usage: tools/store_records.sh [round directory, default r3]
set -e
cd "$(dirname "$0")/.."
python3 tools/merge_counters.py --into profiles/counters.json gpurun_out/counters/config2.json gpurun_out/counters/config3.json gpurun_out/counters/config4.json gpurun_out/counters/config5.json
r=profiles/${1:-r3}
mkdir -p $r
cp gpurun_out/final/bench.json $r/40_final_bench.json
cp gpurun_out/final/bench_under_rocprof.json $r/40_final_bench_under_rocprof.json
cp gpurun_out/final/kt/*kernel_stats.csv $r/40_final_kernel_stats.csv
cp gpurun_out/final/timeline.txt $r/41_launch_timeline.txt
cp gpurun_out/final/context_timing.json $r/40_context_timing.json
cp gpurun_out/final/rank_overhead.jsonl $r/42_rank_overhead.jsonl
cp gpurun_out/final/independent_seed_psnr.json $r/43_independent_seed_psnr.json
cp gpurun_out/final/path_step_bench.jsonl $r/44_path_step_bench.jsonl
cp gpurun_out/final/fullsize_parity.txt $r/45_fullsize_parity.txt
grep '^{' gpurun_out/final/rehearsal_2ranks_fullsize.json > $r/46_rehearsal_2ranks_fullsize.json
[ -f gpurun_out/final/sections_c3.json ] && cp gpurun_out/final/sections_c3.json $r/40_section_counts_config3.json
cp gpurun_out/counters/config3.json $r/40_final_counters_config3.json
for c in 2 4 5; do
  cp gpurun_out/configs/bench_c$c.json $r/10_config${c}_bench.json
  cp gpurun_out/configs/kt_c$c/*kernel_stats.csv $r/10_config${c}_kernel_stats.csv
  cp gpurun_out/counters/config$c.json $r/10_config${c}_counters.json
done
cp gpurun_out/configs/bench_c5_norefill.json $r/10_config5_compaction_off_bench.json
echo stored in $r
