"""Config-4 measurement: all-pairs nearest neighbour over N x 128 fp16 unit rows
(N x N never materialised).  Prints one JSON line with the MFMA roofline of
k_pairwise (2·N·M·128 FLOP / kernel time / 2.5 PFLOP/s).

    python tools/bench_distance.py --rows 1000000 --metric l2
"""
from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ginfinity_amd import distance, synthetic  # noqa: E402


def main() -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument("--rows", type=int, default=1_000_000)
    parser.add_argument("--metric", default="l2")
    parser.add_argument("--repeats", type=int, default=3)
    args = parser.parse_args()
    rows = torch.from_numpy(synthetic.unit_rows(0, args.rows)).cuda()
    distance.nearest(rows[:4096], rows[:4096], metric=args.metric)      # warm
    torch.cuda.synchronize()
    times = []
    for _ in range(args.repeats):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        values, indices = distance.nearest(rows, metric=args.metric, exclude_self=True)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) * 1e-3)
    best = min(times)
    flops = 2.0 * args.rows * args.rows * 128
    print(json.dumps({
        "metric": "all-pairs nearest over N x 128 fp16 embeddings",
        "rows": args.rows, "distance": args.metric, "seconds": best,
        "pairs_per_s": args.rows * args.rows / best,
        "roofline": {"bound": "mfma", "achieved": flops / best / 1e12, "peak": 2500.0,
                     "unit": "TFLOP/s", "frac": flops / best / 1e12 / 2500.0,
                     # what a pure v_mfma_f32_32x32x16_f16 loop (no memory) sustains on this
                     # part with random-mantissa operands: tools/mfma_peak.hip,
                     # profiles/r01c_mfma_peak.txt (2,433 with small-integer operands)
                     "sustained_mfma_measured": 1780.0,
                     "frac_of_sustained": flops / best / 1e12 / 1780.0},
        "all_seconds": times,
        "sample": {"value0": float(values[0]), "index0": int(indices[0])}}))


if __name__ == "__main__":
    main()
