#!/bin/bash
# Build the stand-alone measurement / probe programs (gfx950).  Run after
# `python -m ginfinity_amd.build` (and `--stamps` for the diagnostic library); the
# binaries are git-ignored but travel to the GPU box with the gpurun snapshot.
set -e
cd "$(dirname "$0")/.."
RP='-Wl,-rpath,$ORIGIN/../ginfinity_amd/csrc'
hipcc -O2 tools/gfy_bench.cpp -Iinclude -Lginfinity_amd/csrc -lgfy "$RP" -o tools/gfy_bench
if [ -f ginfinity_amd/csrc/libgfy_stamps.so ]; then
  hipcc -O2 -DGFY_STAMPS tools/gfy_bench.cpp -Iinclude -Lginfinity_amd/csrc -lgfy_stamps "$RP" -o tools/gfy_bench_stamps
fi
for p in clockcheck probe dma_probe dma_rate coherence_probe mfma_peak isa_semantics valu_rate overlap_probe issue_probe; do
  hipcc -O3 --offload-arch=gfx950 tools/$p.hip -o tools/$p
done
# the layer kernels' own pipelines on synthetic operands (includes csrc/gine_f16.hip: same flags)
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function \
  tools/mlp_probe.hip -o tools/mlp_probe
