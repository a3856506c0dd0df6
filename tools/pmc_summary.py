"""Condense the rocprofv3 --pmc passes of tools/profile_round.sh into the JSON files kept
under profiles/ (mean counter value per dispatch and kernel).

    python tools/pmc_summary.py r01c
"""
import collections
import csv
import json
import re
import sys
from pathlib import Path

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = Path(__file__).resolve().parent.parent
src = root / "gpurun_out" / tag
sys.path.insert(0, str(root))
from bench import kernel_source_sha16  # noqa: E402  (bench.py reports `traffic` only while it matches)


def short(name: str) -> str:
    m = re.search(r"(k_[a-z_0-9]+)(I[A-Za-z0-9_]*E)?", name)
    if not m:
        return name[:40]
    base = m.group(1)
    if base == "k_gine_layer_f16" and re.search(r"k_gine_layer_f16ILb[01]ELb1E", name):
        base = "k_gine_layer_f16<+head>"          # <kResidual, kHead>
    if base in ("k_gine_layer_q", "k_gine_layer_w", "k_gine_layer_x") and re.search(
            base + r"ILb[01]ELb[01]ELb1E", name):
        base += "<+head>"                         # <kResidual, kTap, kHead>
    return base


def means(path: Path):
    """Mean counter value per dispatch and kernel — over the dispatches of the kernel's LARGEST
    grid only: a warm-up on a few rows (tools/bench_distance.py runs k_pairwise on 4,096 rows
    before the 1M x 1M search) is another workload, and averaging it in halved every k_pairwise
    counter of round 3's file."""
    rows = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        rows[short(row["Kernel_Name"])].append(row)
    out = {}
    for kernel, dispatches in rows.items():
        grid_of = lambda r: int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)
        largest = max(grid_of(r) for r in dispatches)
        kept = collections.defaultdict(list)
        for r in dispatches:
            if grid_of(r) == largest:
                kept[r["Counter_Name"]].append(float(r["Counter_Value"]))
        out[kernel] = {c: sum(v) / len(v) for c, v in kept.items()}
        out[kernel]["grid_size"] = largest
        out[kernel]["dispatches_averaged"] = max(len(v) for v in kept.values())
    return out


traffic = collections.defaultdict(dict)
for sub in sorted(src.glob("pmc_*")):
    if sub.is_dir():
        for kernel, counters in means(sub / "pmc_counter_collection.csv").items():
            traffic[kernel].update(counters)
for kernel, c in traffic.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        c["hbm_bytes_per_launch"] = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
json.dump({
    "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_* (three separate passes, "
           "--kernel-trace only) -- ./tools/gfy_bench 240000 20 (tools/profile_round.sh); mean per "
           "dispatch; workload = 240,000 nodes / 1,200,000 edges in one launch sequence = four "
           "60,000-node shards' worth (what a batch of 4 launches)",
    "nodes_per_launch": 240000,
    "units": "FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them",
    "kernel_source_sha16": kernel_source_sha16(),
    "correction": "gfx950: FETCH_SIZE reports half the bytes of wide (16 B/lane) coalesced reads "
                  "(MI355X_MICROARCH.md, HBM section) -> hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024",
    "kernels": traffic}, open(root / "profiles" / f"{tag}_traffic_pmc.json", "w"), indent=1)

sq = collections.defaultdict(dict)
for sub in ("sq_a", "sq_b", "sq_c", "sq_a_k5", "sq_c_k5", "sq_a_k3", "sq_c_k3"):
    path = src / sub / "sq_counter_collection.csv"
    if path.exists():
        for kernel, counters in means(path).items():
            sq[kernel].update(counters)
SIMDS = 1024
for kernel, c in sq.items():
    # shares of the launch's cycles per SIMD.  The launch's length in shader cycles is taken as
    # the mean wave lifetime (SQ_WAVE_CYCLES counts quad-cycles per wave; a layer launch's waves
    # live from its start to its end, within a few per cent), the busy counters are cycles
    # summed over the chip's 1,024 SIMDs.
    if c.get("SQ_WAVES") and c.get("SQ_WAVE_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        lifetime = 4.0 * c["SQ_WAVE_CYCLES"] / c["SQ_WAVES"]
        c["wave_lifetime_cycles"] = lifetime
        c["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / SIMDS / lifetime
        if "SQ_VALU_MFMA_COEXEC_CYCLES" in c:
            c["coexec_frac"] = c["SQ_VALU_MFMA_COEXEC_CYCLES"] / SIMDS / lifetime
        if "SQ_ACTIVE_INST_VALU" in c:   # quad-cycles: one per vector instruction (incl. MFMA issue)
            c["valu_active_frac"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / SIMDS / lifetime
json.dump({
    "how": "rocprofv3 --pmc passes of <= 8 SQ counters each over ./tools/gfy_bench 240000 20 "
           "(tools/profile_round.sh: sq_a, sq_b, sq_c for the default kernel, _k5 / _k3 with "
           "GFY_BENCH_LAYER_KERNEL=5 / 3); mean per dispatch, summed over the chip "
           "(SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave, "
           "SQ_VALU_MFMA_BUSY_CYCLES, SQ_VALU_MFMA_COEXEC_CYCLES and SQ_LDS_* count cycles); "
           "*_frac: shares of the launch's cycles per SIMD (1,024 SIMDs, launch length = mean "
           "wave lifetime)",
    "kernel_source_sha16": kernel_source_sha16(),
    "kernels": sq}, open(root / "profiles" / f"{tag}_layer_sq_pmc.json", "w"), indent=1)
pairwise = src / "sq_pairwise" / "sq_counter_collection.csv"
if pairwise.exists():
    json.dump({
        "how": "rocprofv3 --pmc (8 SQ counters, --kernel-trace only) -- python3 "
               "tools/bench_distance.py --repeats 1: nearest other row of 1,000,000 x 128 fp16 "
               "rows; mean per dispatch, summed over the chip; HBM traffic of the same launches "
               "is in the traffic file",
        "kernels": {k: v for k, v in means(pairwise).items() if k.startswith("k_")}},
        open(root / "profiles" / f"{tag}_pairwise_sq_pmc.json", "w"), indent=1)
for kernel in sorted(traffic):
    c = traffic[kernel]
    print(f"{kernel:28s} FETCH {c.get('FETCH_SIZE', 0):9.1f} KiB  WRITE {c.get('WRITE_SIZE', 0):9.1f} KiB  "
          f"-> {c.get('hbm_bytes_per_launch', 0) / 1e6:7.2f} MB/launch  "
          f"TCC hit {c.get('TCC_HIT_sum', 0):9.0f} miss {c.get('TCC_MISS_sum', 0):9.0f}")
