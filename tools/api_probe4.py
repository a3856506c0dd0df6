"""encode_graphs on the rouskin shard, call after call: page-locked results (default) and pageable
ones, alternating blocks of 8 calls (diagnostic: variance and order effects)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from ginfinity_amd import Ginfinity, GraphBuilder, read_rna_table

records = read_rna_table(ROOT / "tests" / "golden" / "rouskin_sample_6k.tsv")
shard = GraphBuilder().build_shard(records)
enc = Ginfinity.load("cuda", allow_nondeterministic_cuda=True)
out = None
for mode in (None, False, None, False, None):
    enc.pinned_outputs = mode
    times = []
    for _ in range(8):
        out = None
        a = time.perf_counter()
        out = enc.encode_graphs(shard)
        times.append((time.perf_counter() - a) * 1e3)
    print("page-locked" if mode is None else "pageable   ", " ".join(f"{t:6.2f}" for t in times), flush=True)
