#!/bin/bash
# Copy the summaries of `tools/profile_round.sh <tag>` (gpurun_out/<tag>/) into profiles/:
#   python tools/pmc_summary.py <tag> && bash tools/collect_profiles.sh <tag>
set -e
T=${1:-r05}
cd "$(dirname "$0")/.."
S=gpurun_out/$T
cp $S/bench_line.json profiles/${T}_bench_line.json
cp $S/bench_1stream_line.json profiles/${T}_bench_1stream_line.json
cp $S/bench_single_line.json profiles/${T}_bench_single_shard_line.json
cp $S/bench_driver_line.json profiles/${T}_bench_driver_args_line.json
cp $S/bench_1stream_line_profiled.json profiles/${T}_bench_1stream_line_profiled.json
cp $S/bench_2stream_line_profiled.json profiles/${T}_bench_2streams_line_profiled.json
cp $S/bench_single_line_profiled.json profiles/${T}_bench_single_shard_line_profiled.json
cp $S/kt1/kt_kernel_stats.csv profiles/${T}_bench_1stream_kernel_stats.csv
cp $S/kt2/kt_kernel_stats.csv profiles/${T}_bench_2streams_kernel_stats.csv
cp $S/kt_single/kt_kernel_stats.csv profiles/${T}_bench_single_shard_kernel_stats.csv
cp $S/gfy_bench_stamps_60k.txt profiles/${T}_layer_stamps_single_shard.txt
cp $S/kt_pairwise/kt_kernel_stats.csv profiles/${T}_pairwise_kernel_stats.csv
cp $S/gfy_bench.txt profiles/${T}_gfy_bench.txt
cp $S/gfy_bench_stamps.txt profiles/${T}_layer_stamps.txt
cp $S/api_bench.json profiles/${T}_api_bench.json
cp $S/distance_bench.json profiles/${T}_distance_bench.json
for f in mlp_probe issue_probe; do [ -f $S/$f.txt ] && cp $S/$f.txt profiles/${T}_$f.txt; done
[ -f $S/gfy_bench_stamps_k5.txt ] && cp $S/gfy_bench_stamps_k5.txt profiles/${T}_layer_stamps_kernel5.txt
[ -f $S/distance_resident.json ] && cp $S/distance_resident.json profiles/${T}_distance_by_b_size.json
if ls $S/gfy_bench_k*.txt > /dev/null 2>&1; then   # the layer kernels on ONE box, back to back
  { echo "tools/gfy_bench 240000 100 with GFY_BENCH_LAYER_KERNEL = 4 (windowed, default), 5 (three workgroups per CU),";
    echo "3 (persistent rounds), 1 (one round): us per launch by HIP events (set-up | layers 1-3 | last + head | -), two calls in flight";
    for K in 4 5 3 1; do echo "kernel $K: $(grep per-kernel $S/gfy_bench_k$K.txt) | $(grep '2 streams' $S/gfy_bench_k$K.txt)"; done; } \
    > profiles/${T}_layer_kernels_same_box.txt
fi
[ -f gpurun_out/parity_margins.json ] && cp gpurun_out/parity_margins.json profiles/${T}_parity_margins.json
echo "profiles/${T}_* refreshed"
