"""Where the time of encode_graphs goes on the host side (diagnostic)."""
import sys, time, threading
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch
from ginfinity_amd import Ginfinity, GraphBuilder, read_rna_table
from ginfinity_amd import api

records = read_rna_table(ROOT / "tests" / "golden" / "rouskin_sample_6k.tsv")
shard = GraphBuilder().build_shard(records)
enc = Ginfinity.load("cuda", allow_nondeterministic_cuda=True)
enc.encode_graphs(shard)
enc.encode_graphs(shard)

# 1. raw costs
n = shard.node_count
t = time.perf_counter(); block = np.empty((n, 128), np.float16); t1 = time.perf_counter() - t
def touch(a):
    a[::2048 // a.itemsize // 128 or 1] = 0
t = time.perf_counter(); block[:] = 0; t2 = time.perf_counter() - t
src = np.ones((n, 128), np.float16)
t = time.perf_counter(); np.copyto(block, src); t3 = time.perf_counter() - t
print(f"np.empty {t1*1e3:.2f} ms, first touch (1 thread) {t2*1e3:.2f} ms, warm memcpy 230 MB {t3*1e3:.2f} ms")
fresh = np.empty((n, 128), np.float16)
parts = np.array_split(np.arange(n), 4)
def cp(lo, hi): np.copyto(fresh[lo:hi], src[lo:hi])
ths = [threading.Thread(target=cp, args=(p[0], p[-1] + 1)) for p in parts]
t = time.perf_counter(); [x.start() for x in ths]; [x.join() for x in ths]; t4 = time.perf_counter() - t
print(f"cold memcpy into fresh block, 4 threads: {t4*1e3:.2f} ms")
ths = [threading.Thread(target=cp, args=(p[0], p[-1] + 1)) for p in parts]
t = time.perf_counter(); [x.start() for x in ths]; [x.join() for x in ths]; t5 = time.perf_counter() - t
print(f"warm memcpy, 4 threads: {t5*1e3:.2f} ms")
pin = torch.empty(n * 256, dtype=torch.uint8, pin_memory=True)
dev = torch.empty(n * 256, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize(); t = time.perf_counter(); pin.copy_(dev, non_blocking=True); torch.cuda.synchronize(); t6 = time.perf_counter() - t
print(f"D2H 230 MB into pinned: {t6*1e3:.2f} ms = {n*256/t6/1e9:.1f} GB/s")
t = time.perf_counter(); x = shard.edge_index[:, :300000] - np.int32(5); x.min(); x.max(); t7 = time.perf_counter() - t
print(f"rebase + min/max of one 300k-edge window: {t7*1e3:.2f} ms")

# 2. the call, traced
trace = []
orig_run = api._Downloader._run
def traced(self, block, ready, finish, destination):
    a = time.perf_counter(); r = orig_run(self, block, ready, finish, destination); trace.append(("copy", a, time.perf_counter())); return r
api._Downloader._run = traced
orig_pack = api._Uploader.pack
def tpack(self, slot, arrays):
    a = time.perf_counter(); r = orig_pack(self, slot, arrays); trace.append(("pack", a, time.perf_counter())); return r
api._Uploader.pack = tpack
import gc
out = None
for _ in range(3):
    trace.clear()
    ta = time.perf_counter(); out = None; gc.collect(); tb = time.perf_counter()
    t0 = time.perf_counter(); out = enc.encode_graphs(shard); t1 = time.perf_counter()
print(f"freeing the previous result {1e3*(tb-ta):.2f} ms; encode_graphs {1e3*(t1-t0):.2f} ms")

for kind in ("pack", "copy"):
    ev = [(a - t0, b - t0) for k, a, b in trace if k == kind]
    print(kind, " ".join(f"{a*1e3:.1f}-{b*1e3:.1f}" for a, b in ev))

# 3. what an asynchronous H2D of 4.4 MB from pinned memory costs the calling thread
stage = torch.empty(6 << 20, dtype=torch.uint8, pin_memory=True)
total = 4_400_000
for label in ("idle stream", "busy stream"):
    times = {"empty": 0.0, "copy": 0.0, "event": 0.0}
    for _ in range(20):
        if label == "busy stream":
            enc.encode_graphs(shard.slice(0, 400))
        a = time.perf_counter(); dev_buf = torch.empty(total, dtype=torch.uint8, device="cuda")
        b = time.perf_counter(); dev_buf[:total].copy_(stage[:total], non_blocking=True)
        c = time.perf_counter(); ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream())
        d = time.perf_counter()
        times["empty"] += b - a; times["copy"] += c - b; times["event"] += d - c
        torch.cuda.synchronize()
    print(label, {k: f"{v / 20 * 1e6:.0f} us" for k, v in times.items()})
