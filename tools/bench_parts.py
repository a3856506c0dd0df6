"""What do the CSR build and the setup launch cost with four shards in flight?  Times the
bench.py loop with (a) CSR build + encode, (b) encode only (CSR reused)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ginfinity_amd import Ginfinity, synthetic
from ginfinity_amd import _native as native

NODES, lanes, steps = 60000, 4, 1000
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
engines = [Ginfinity.load("cuda:0", allow_nondeterministic_cuda=True)._engine for _ in range(lanes)]
streams = [torch.cuda.Stream(device=dev) for _ in range(lanes)]
shards = [synthetic.roofline_shard(i) for i in range(lanes)]
inputs = [(torch.from_numpy(s.node_features).to(dev), torch.from_numpy(s.edge_index).to(dev),
           torch.from_numpy(s.edge_types).to(dev)) for s in shards]
outs = [torch.empty((NODES, 128), dtype=torch.float16, device=dev) for _ in range(lanes)]
full = [engines[l].prepare_step(*inputs[l], outs[l]) for l in range(lanes)]
csrs = [engines[l].build_csr(inputs[l][1], inputs[l][2], NODES) for l in range(lanes)]
handles = [s.cuda_stream for s in streams]

def run(fn, label):
    for i in range(100): fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps): fn(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{label}: {1e6*dt/steps:.1f} us/step -> {steps*NODES/dt/1e6:.1f} M nodes/s")

run(lambda i: full[i % lanes](handles[i % lanes]), "csr + encode")
def enc_only(i):
    l = i % lanes
    with torch.cuda.stream(streams[l]):
        engines[l].encode(inputs[l][0], csrs[l], out=outs[l])
run(enc_only, "encode only ")
