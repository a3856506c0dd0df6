"""Where the time of encode_graphs(pinned_outputs=True) goes (diagnostic)."""
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
enc.encode_graphs(shard)
enc.encode_graphs(shard)

trace = []
t0 = [0.0]
def wrap(cls, name, label):
    orig = getattr(cls, name)
    def traced(self, *a, **k):
        s = time.perf_counter(); r = orig(self, *a, **k); trace.append((label, s - t0[0], time.perf_counter() - t0[0])); return r
    setattr(cls, name, traced)
wrap(api._Uploader, "pack", "pack")
wrap(api._Uploader, "send", "send")
marks = []
wrap(api._DirectDownloader, "submit", "d2h-enqueue")
wrap(api._DirectDownloader, "_pump", "pump")
wrap(api._DirectDownloader._Landed, "result", "landed")
orig_landing = api.Ginfinity._landing
def landing(self, *a, **k):
    s = time.perf_counter(); r = orig_landing(self, *a, **k); trace.append(("landing", s - t0[0], time.perf_counter() - t0[0])); return r
api.Ginfinity._landing = landing
orig_enc = enc._engine.encode_coo
def enc_coo(*a, **k):
    s = time.perf_counter(); r = orig_enc(*a, **k); trace.append(("encode_coo", s - t0[0], time.perf_counter() - t0[0])); return r
enc._engine.encode_coo = enc_coo
out = None
import os
for rep in range(int(os.environ.get("GFY_PROBE_CALLS", "3"))):
    out = None
    trace.clear()
    marks.clear()
    torch.cuda.synchronize()
    t0[0] = time.perf_counter()
    out = enc.encode_graphs(shard)
    total = time.perf_counter() - t0[0]
    torch.cuda.synchronize()
    copies = [0.0]
    last = lambda label: max((b for l, a, b in trace if l == label), default=0.0) * 1e3
    first = lambda label: min((a for l, a, b in trace if l == label), default=0.0) * 1e3
    print(f"call {rep:2d}: {total*1e3:6.2f} ms | packs end {last('pack'):5.2f} sends end {last('send'):5.2f} "
          f"enqueues end {last('d2h-enqueue'):5.2f} first landed {first('landed'):5.2f} last landed {last('landed'):5.2f} | "
          "landed ends: " + " ".join(f"{b*1e3:.1f}" for l, a, b in trace if l == "landed"), flush=True)
