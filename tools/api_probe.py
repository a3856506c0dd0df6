"""A/B of the two micro-batch paths of encode_graphs in ONE process (host-side costs
vary a lot between boxes)."""
import sys, time, numpy as np, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ginfinity_amd import Ginfinity, GraphBuilder, read_rna_table
from ginfinity_amd.api import microbatch_bounds
recs = read_rna_table(Path(__file__).resolve().parents[1] / "tests/golden/rouskin_sample_6k.tsv")
shard = GraphBuilder().build_shard(recs)
enc = Ginfinity.load("cuda", allow_nondeterministic_cuda=True)
enc.encode_graphs(shard.slice(0, 50))
bounds = microbatch_bounds(shard.lengths, shard.edge_counts, 60000, 300000)
def old():
    out = []
    for a, b in bounds:
        out.extend(enc._run_graph_shard(shard.slice(a, b), np.dtype(np.float16)))
    return out
def new():
    return enc.encode_graphs(shard)
for rnd in range(4):
    for name, f in (("slice+validate per micro-batch", old), ("direct array cuts", new)):
        t = time.perf_counter(); r = f(); dt = time.perf_counter() - t
        print(f"round {rnd} {name:32s} {dt*1e3:7.2f} ms  {897588/dt/1e6:6.1f} M nodes/s")
