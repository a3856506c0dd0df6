"""Does D2H bandwidth depend on the NUMA node the page-locked block was allocated from?"""
import os, glob, time
import torch

dev = torch.device("cuda:0")
src = torch.empty(230 << 20, dtype=torch.uint8, device=dev)
print("cpus", os.cpu_count(), "affinity now", len(os.sched_getaffinity(0)))
nodes = {}
for path in sorted(glob.glob("/sys/devices/system/node/node*/cpulist")):
    node = int(path.split("node")[-1].split("/")[0])
    cpus = set()
    for part in open(path).read().strip().split(","):
        if "-" in part:
            a, b = part.split("-"); cpus.update(range(int(a), int(b) + 1))
        elif part:
            cpus.add(int(part))
    nodes[node] = cpus
print("numa nodes:", {k: len(v) for k, v in nodes.items()})
for path in glob.glob("/sys/class/drm/card*/device/numa_node"):
    print(path, open(path).read().strip())
for path in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/io_links/*/properties"):
    pass
allowed = os.sched_getaffinity(0)
for node, cpus in nodes.items():
    usable = cpus & allowed
    if not usable:
        print("node", node, "no usable cpus"); continue
    os.sched_setaffinity(0, usable)
    block = torch.empty(230 << 20, dtype=torch.uint8, pin_memory=True)
    block.zero_()
    os.sched_setaffinity(0, allowed)
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize(); t = time.perf_counter()
        block.copy_(src, non_blocking=True); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    print(f"pinned block allocated on node {node}: D2H {230*1.048576/best/1e3:.1f} GB/s", flush=True)
    del block
    torch._C._host_emptyCache() if hasattr(torch._C, "_host_emptyCache") else None
