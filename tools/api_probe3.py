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
enc = Ginfinity.load("cuda", allow_nondeterministic_cuda=True, pinned_outputs=True)
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
orig_submit = api._DirectDownloader.submit
def submit(self, block, ready, destination):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    st = self._streams[self._turn]
    st.wait_event(ready)
    a.record(st)
    r = orig_submit(self, block, ready, destination)
    b.record(st)
    marks.append((a, b, block.data_ptr(), destination.data_ptr()))
    return r
api._DirectDownloader.submit = submit
wrap(api._DirectDownloader, "submit", "d2h-enqueue")
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
for rep in range(3):
    out = None
    trace.clear()
    marks.clear()
    torch.cuda.synchronize()
    t0[0] = time.perf_counter()
    out = enc.encode_graphs(shard)
    total = time.perf_counter() - t0[0]
    print(f"--- call {rep}: {total*1e3:.2f} ms")
    torch.cuda.synchronize()
    print("gpu-side copy ms:", " ".join(f"{a.elapsed_time(b):.2f}" for a, b, _s, _d in marks))
    print("gpu-side start of copy i relative to copy 0 start:", " ".join(f"{marks[0][0].elapsed_time(a):.2f}" for a, b, _s, _d in marks))
    print("source blocks:", " ".join(hex(s_)[-9:] for a, b, s_, _d in marks))
    for label in ("landing", "pack", "send", "encode_coo", "d2h-enqueue", "landed"):
        rows = [(a, b) for l, a, b in trace if l == label]
        if rows:
            print(f"{label:12s} n={len(rows):3d} sum {sum(b-a for a,b in rows)*1e3:6.2f} ms  first starts {rows[0][0]*1e3:6.2f}  last ends {rows[-1][1]*1e3:6.2f}"
                  + ("   each end: " + " ".join(f"{b*1e3:.1f}" for a, b in rows) if label in ("landed", "send") else ""))
