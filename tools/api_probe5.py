"""When do the micro-batches of encode_graphs finish on the GPU, and when do their copies land?
(diagnostic for the 5.4 ms / 7.3 ms modes of the call)"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch
from ginfinity_amd import Ginfinity, GraphBuilder, read_rna_table
from ginfinity_amd import api

records = read_rna_table(ROOT / "tests" / "golden" / "rouskin_sample_6k.tsv")
shard = GraphBuilder().build_shard(records)
enc = Ginfinity.load("cuda", allow_nondeterministic_cuda=True)
enc.encode_graphs(shard); enc.encode_graphs(shard)

_Event = torch.cuda.Event
class TimedEvent(_Event):
    def __new__(cls, *a, **k):
        return super().__new__(cls, enable_timing=True)
torch.cuda.Event = TimedEvent
readies, landed = [], []
orig_submit = api._DirectDownloader.submit
def submit(self, block, ready, destination):
    readies.append(ready)
    job = orig_submit(self, block, ready, destination)
    landed.append(job)
    return job
api._DirectDownloader.submit = submit
out = None
for rep in range(10):
    out = None
    readies.clear(); landed.clear()
    torch.cuda.synchronize()
    start = TimedEvent(); start.record()
    t0 = time.perf_counter()
    out = enc.encode_graphs(shard)
    total = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    print(f"call {rep}: {total:5.2f} ms | kernels of micro-batch i done at (ms): "
          + " ".join(f"{start.elapsed_time(r):.1f}" for r in readies)
          + " | copies done at: " + " ".join(f"{start.elapsed_time(j._done):.1f}" for j in landed), flush=True)
