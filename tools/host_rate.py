"""How fast can the Python host enqueue steps (CSR build + encode) without waiting for the
GPU?  If this is close to the GPU's time per step, bench.py is host-bound."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ginfinity_amd import Ginfinity, synthetic

dev = torch.device("cuda", 0)
enc = Ginfinity.load("cuda:0", allow_nondeterministic_cuda=True)._engine
s = synthetic.roofline_shard(0)
x = torch.from_numpy(s.node_features).to(dev)
ei = torch.from_numpy(s.edge_index).to(dev)
et = torch.from_numpy(s.edge_types).to(dev)
out = torch.empty((60000, 128), dtype=torch.float16, device=dev)
csr = enc.build_csr(ei, et, 60000)
enc.encode(x, csr, out=out)
torch.cuda.synchronize()
for steps in (200, 2000):
    t0 = time.perf_counter()
    for _ in range(steps):
        csr = enc.build_csr(ei, et, 60000, out=csr)
        enc.encode(x, csr, out=out)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{steps} steps: host enqueue {1e6*(t1-t0)/steps:.1f} us/step, with GPU drain {1e6*(t2-t0)/steps:.1f} us/step")
