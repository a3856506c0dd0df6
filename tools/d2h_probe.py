"""Chunked D2H into one pinned block: per-copy durations (diagnostic for api._DirectDownloader)."""
import sys, time
import torch

dev = torch.device("cuda:0")
rows, width, chunks = 897588, 128, 15
per = rows // chunks
landing = torch.empty((rows, width), dtype=torch.float16, pin_memory=True)
blocks = [torch.randn((per, width), device=dev).half() for _ in range(chunks)]
stream = torch.cuda.Stream(device=dev)
busy = torch.randn((4096, 4096), device=dev)


up_stream = torch.cuda.Stream(device=dev)
up_src = [torch.empty(4 << 20, dtype=torch.uint8, pin_memory=True) for _ in range(15)]
up_dst = [torch.empty(4 << 20, dtype=torch.uint8, device=dev) for _ in range(15)]


def run(label, compute_ms=0.0, split=1, uploads=False):
    torch.cuda.synchronize()
    marks = []
    t0 = time.perf_counter()
    if uploads:
        with torch.cuda.stream(up_stream):
            for _ in range(4):
                for a, b in zip(up_src, up_dst):
                    b.copy_(a, non_blocking=True)
    with torch.cuda.stream(stream):
        for i, block in enumerate(blocks):
            a = torch.cuda.Event(enable_timing=True); a.record(stream)
            step = per // split
            for k in range(split):
                lo = k * step
                hi = per if k == split - 1 else lo + step
                landing[i * per + lo:i * per + hi].copy_(block[lo:hi], non_blocking=True)
            b = torch.cuda.Event(enable_timing=True); b.record(stream)
            marks.append((a, b))
    if compute_ms:
        until = time.perf_counter() + compute_ms * 1e-3
        while time.perf_counter() < until:
            torch.mm(busy, busy)
    stream.synchronize()
    total = time.perf_counter() - t0
    torch.cuda.synchronize()
    each = [a.elapsed_time(b) for a, b in marks]
    print(f"{label:34s} total {total*1e3:6.2f} ms = {rows*256/total/1e9:5.1f} GB/s   per copy ms: "
          + " ".join(f"{x:.2f}" for x in each))


for rep in range(3):
    run("idle GPU, 15 copies")
for rep in range(3):
    run("compute for the first 3 ms", compute_ms=3.0)
for rep in range(2):
    run("compute throughout (8 ms)", compute_ms=8.0)
for rep in range(2):
    run("idle, each copy in 4 pieces", split=4)
for rep in range(2):
    run("compute 3 ms, 4 pieces", compute_ms=3.0, split=4)
for rep in range(4):
    run("with 240 MB of H2D in 4 MB copies", uploads=True)
for rep in range(4):
    run("H2D + compute 3 ms", uploads=True, compute_ms=3.0)
one = torch.empty((rows, width), dtype=torch.float16, device=dev)
torch.cuda.synchronize(); t = time.perf_counter()
landing.copy_(one, non_blocking=True); torch.cuda.synchronize()
t = time.perf_counter() - t
print(f"one copy of {rows*256/1e6:.0f} MB: {t*1e3:.2f} ms = {rows*256/t/1e9:.1f} GB/s")
