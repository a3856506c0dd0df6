"""Does k_pairwise's fabric traffic (290 GB per 1M x 1M search: 3,907 workgroups each streaming
all of B with a 71 % L2 hit rate, profiles/r04_traffic_pmc.json) cost it clock or pipe time?
Timing only: the same 1M query rows against B slices of decreasing size — 1M rows (256 MB: HBM /
Infinity Cache), 65,536 rows (16 MB: Infinity Cache, half of it per XCD's L2s), 8,192 rows (2 MB:
every XCD's L2 holds it) — every search repeated so that each timed leg does about the work of
one full search.  TFLOP/s = 2 N M 128 / time.  If the small-B legs are no faster, the traffic is
not what limits the search.

    python tools/bench_distance_resident.py
"""
from __future__ import annotations

import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ginfinity_amd import distance, synthetic  # noqa: E402


def main() -> None:
    rows = torch.from_numpy(synthetic.unit_rows(0, 1_000_000)).cuda()
    work = distance.NearestWorkspace()
    distance.nearest(rows[:4096], rows[:4096], workspace=work)      # warm
    distance.nearest(rows, rows[:65536], workspace=work)
    torch.cuda.synchronize()
    out = {}
    for m in (1_000_000, 262_144, 65_536, 8_192):
        repeats = max(1, 1_000_000 // m)
        best = None
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(repeats):
                distance.nearest(rows, rows[:m], workspace=work)
            e1.record()
            torch.cuda.synchronize()
            seconds = e0.elapsed_time(e1) * 1e-3
            best = seconds if best is None or seconds < best else best
        flops = 2.0 * 1_000_000 * m * repeats * 128
        out[str(m)] = {"b_rows": m, "b_megabytes": m * 256 / 1e6, "searches": repeats,
                       "seconds": best, "tflops": flops / best / 1e12,
                       "frac_of_2500": flops / best / 1e12 / 2500.0}
    print(json.dumps({"what": "nearest(1M query rows, B slice) by the size of B; L2 metric",
                      "legs": out}))


if __name__ == "__main__":
    main()
