#!/bin/bash
# Copy the summaries of `tools/profile_round.sh <tag>` (gpurun_out/<tag>/) into profiles/:
#   python tools/pmc_summary.py <tag> && bash tools/collect_profiles.sh <tag>
set -e
T=${1:-r03}
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
[ -f gpurun_out/parity_margins.json ] && cp gpurun_out/parity_margins.json profiles/${T}_parity_margins.json
echo "profiles/${T}_* refreshed"
