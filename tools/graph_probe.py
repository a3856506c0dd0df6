"""Does replaying one step (CSR build + encode, 10 launches) as a captured HIP graph shorten
the step?  One stream, 60k-node shard; compares stream launches with graph replays."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ginfinity_amd import Ginfinity, synthetic

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
enc = Ginfinity.load("cuda:0", allow_nondeterministic_cuda=True)._engine
s = synthetic.roofline_shard(0)
x = torch.from_numpy(s.node_features).to(dev)
ei = torch.from_numpy(s.edge_index).to(dev)
et = torch.from_numpy(s.edge_types).to(dev)
out = torch.empty((60000, 128), dtype=torch.float16, device=dev)
step = enc.prepare_step(x, ei, et, out)
stream = torch.cuda.Stream(device=dev)
handle = stream.cuda_stream
for _ in range(20): step(handle)
torch.cuda.synchronize()
want = out.clone()

def timed(fn, n=1000):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n

print(f"stream launches: {timed(lambda: step(handle)):.1f} us/step")
graph = torch.cuda.CUDAGraph()
out.zero_()
with torch.cuda.graph(graph, stream=stream):
    step(stream.cuda_stream)
torch.cuda.synchronize()
graph.replay(); torch.cuda.synchronize()
print("graph result identical:", bool(torch.equal(out, want)))
print(f"graph replays:   {timed(graph.replay):.1f} us/step")
