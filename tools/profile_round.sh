#!/bin/bash
# Reproduce the committed measurement set of a round on a GPU box:
#   python -m ginfinity_amd.build && python -m ginfinity_amd.build --stamps && bash tools/build_tools.sh
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/profile_round.sh r05 a'   (then ... r05 b)
# Writes gpurun_out/<tag>/…; `python tools/pmc_summary.py <tag>` condenses the counter passes
# and the summaries judged are then copied into profiles/ (see profiles/README.md).
set -o pipefail
TAG=${1:-r05}
PART=${2:-all}     # "a": bench lines, kernel traces, counter passes; "lines": the bench lines only; "b": drivers, stamps, distance, API; "all"
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp

if [ "$PART" != "b" ]; then
# 1. bench lines: the default (batches of 4 shards, 2 in flight), one batch at a time, and one
#    SHARD at a time (single-round launches: the round-2 kernel with the fused head)
timeout -k 10 400 python3 $R/bench.py --steps 960 --warmup 96 > $OUT/bench_line.json 2> $OUT/bench_line.err || exit 1
timeout -k 10 200 python3 $R/bench.py --steps 960 --warmup 96 --streams 1 --no-cpu-baseline --distance-rows 0 > $OUT/bench_1stream_line.json 2>> $OUT/bench_line.err || exit 1
timeout -k 10 200 python3 $R/bench.py --steps 960 --warmup 96 --streams 1 --batch 1 --no-cpu-baseline --distance-rows 0 > $OUT/bench_single_line.json 2>> $OUT/bench_line.err || exit 1
# ... and what the driver runs (20 steps, 5 warm-up): the repeated leg covers >= 10 ms
timeout -k 10 200 python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --distance-rows 0 > $OUT/bench_driver_line.json 2>> $OUT/bench_line.err || exit 1

[ "$PART" = "lines" ] && { echo done; exit 0; }   # "lines": the bench lines only (after pmc_summary.py has refreshed the counter files they quote)
# 2. per-kernel durations of the same commands (kernel trace + stats only)
for S in 1 2; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt$S -o kt --output-format csv -- \
    python3 $R/bench.py --steps 960 --warmup 96 --streams $S --no-cpu-baseline --distance-rows 0 > $OUT/bench_${S}stream_line_profiled.json 2> $OUT/kt$S.err || exit 1
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt_single -o kt --output-format csv -- \
  python3 $R/bench.py --steps 960 --warmup 96 --streams 1 --batch 1 --no-cpu-baseline --distance-rows 0 > $OUT/bench_single_line_profiled.json 2> $OUT/kt_single.err || exit 1

# 3. HBM traffic counters: separate passes, kernel trace only (MI355X_MICROARCH.md, HBM section).
#    240,000 nodes in one graph = four shards' worth: the launches a batch of 4 makes
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  N=$(echo $C | cut -d' ' -f1)
  GFY_BENCH_STREAMS=1 timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_$N -o pmc --output-format csv -- \
    $R/tools/gfy_bench 240000 20 > $OUT/pmc_$N.log 2>&1 || exit 1
done

# 4. SQ counters of the layer kernel (two passes of 8)
GFY_BENCH_STREAMS=1 timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS \
  --kernel-trace -d $OUT/sq_a -o sq --output-format csv -- $R/tools/gfy_bench 240000 20 > $OUT/sq_a.log 2>&1 || exit 1
GFY_BENCH_STREAMS=1 timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM SQ_INSTS_MFMA \
  --kernel-trace -d $OUT/sq_b -o sq --output-format csv -- $R/tools/gfy_bench 240000 20 > $OUT/sq_b.log 2>&1 || exit 1

# 4b. what the occupancy / overlap question is about: cycles in which vector and matrix
#     instructions execute together (MI355X_MICROARCH.md, "Two waves per SIMD", item 9), with the
#     wave count that turns SQ_WAVE_CYCLES into a launch length — for the default kernel, the
#     three-workgroup kernel (5) and the persistent rounds (3)
SQC="SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU"
GFY_BENCH_STREAMS=1 timeout -k 10 200 rocprofv3 --pmc $SQC --kernel-trace -d $OUT/sq_c -o sq --output-format csv -- $R/tools/gfy_bench 240000 20 > $OUT/sq_c.log 2>&1 || exit 1
for K in 5 3; do
  GFY_BENCH_LAYER_KERNEL=$K GFY_BENCH_STREAMS=1 timeout -k 10 200 rocprofv3 --pmc $SQC --kernel-trace -d $OUT/sq_c_k$K -o sq --output-format csv -- $R/tools/gfy_bench 240000 20 > $OUT/sq_c_k$K.log 2>&1 || exit 1
  GFY_BENCH_LAYER_KERNEL=$K GFY_BENCH_STREAMS=1 timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS \
    --kernel-trace -d $OUT/sq_a_k$K -o sq --output-format csv -- $R/tools/gfy_bench 240000 20 > $OUT/sq_a_k$K.log 2>&1 || exit 1
done
# ... the same kernels' durations on this box, back to back (boxes differ by +-5 %)
for K in 4 5 3 1; do
  GFY_BENCH_LAYER_KERNEL=$K timeout -k 10 100 $R/tools/gfy_bench 240000 100 > $OUT/gfy_bench_k$K.txt 2>&1 || exit 1
done
timeout -k 10 120 $R/tools/mlp_probe 200 > $OUT/mlp_probe.txt 2>&1 || exit 1
timeout -k 10 120 $R/tools/issue_probe > $OUT/issue_probe.txt 2>&1 || exit 1
GFY_BENCH_LAYER_KERNEL=5 GFY_BENCH_STREAMS=1 timeout -k 10 100 $R/tools/gfy_bench_stamps 240000 50 > $OUT/gfy_bench_stamps_k5.txt 2>&1 || exit 1

fi
if [ "$PART" != "a" ]; then
# 5. C++ driver (no Python in the loop) and in-kernel phase stamps (diagnostic build)
timeout -k 10 100 $R/tools/gfy_bench 240000 200 > $OUT/gfy_bench.txt 2>&1 || exit 1
GFY_BENCH_STREAMS=1 timeout -k 10 100 $R/tools/gfy_bench_stamps 240000 50 > $OUT/gfy_bench_stamps.txt 2>&1 || exit 1
GFY_BENCH_STREAMS=1 timeout -k 10 100 $R/tools/gfy_bench_stamps 60000 50 > $OUT/gfy_bench_stamps_60k.txt 2>&1 || exit 1

# 6. all-pairs distance (config 4): kernel durations and traffic of k_pairwise
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt_pairwise -o kt --output-format csv -- \
  python3 $R/tools/bench_distance.py > $OUT/distance_bench.json 2> $OUT/distance.err || exit 1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_pairwise_$C -o pmc --output-format csv -- \
    python3 $R/tools/bench_distance.py --repeats 1 > $OUT/pmc_pairwise_$C.log 2>&1 || exit 1
done
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT \
  --kernel-trace -d $OUT/sq_pairwise -o sq --output-format csv -- python3 $R/tools/bench_distance.py --repeats 1 > $OUT/sq_pairwise.log 2>&1 || exit 1

# 7. API level incl. PCIe (config 2) and shard file -> embeddings
timeout -k 10 300 python3 $R/tools/bench_api.py > $OUT/api_bench.json 2> $OUT/api.err || exit 1
fi
echo done
