#!/bin/bash
# the four bench lines of tools/profile_round.sh only (after `pmc_summary.py` has refreshed the
# traffic file from the same kernel sources, so that roofline.traffic is filled in)
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
timeout -k 10 400 python3 $R/bench.py --steps 960 --warmup 96 > $OUT/bench_line.json 2> $OUT/bench_line.err || exit 1
timeout -k 10 200 python3 $R/bench.py --steps 960 --warmup 96 --streams 1 --no-cpu-baseline --distance-rows 0 > $OUT/bench_1stream_line.json 2>> $OUT/bench_line.err || exit 1
timeout -k 10 200 python3 $R/bench.py --steps 960 --warmup 96 --streams 1 --batch 1 --no-cpu-baseline --distance-rows 0 > $OUT/bench_single_line.json 2>> $OUT/bench_line.err || exit 1
timeout -k 10 200 python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --distance-rows 0 > $OUT/bench_driver_line.json 2>> $OUT/bench_line.err || exit 1
echo lines done
