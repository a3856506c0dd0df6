"""Per-basic-block instruction counts of one kernel in a hipcc -S listing.

    python tools/isa_blocks.py layer.s k_gine_layer_f16ILb1ELb0 [first_line last_line]

Prints, for every label (and barrier), the number of VALU / SALU / LDS / VMEM / MFMA
instructions up to the next label, so that the hot path of a phase can be summed by hand.
"""
import re
import sys

path, needle = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and needle in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hi = int(sys.argv[4]) if len(sys.argv) > 4 else 10**9


def flush(name, c, first):
    if sum(c.values()):
        print(f"{first:5d} {name:12s} valu {c['v']:4d} salu {c['s']:4d} lds {c['d']:3d} "
              f"vmem {c['g']:3d} mfma {c['m']:3d} branch {c['b']:2d}")


name, first = "entry", 0
c = dict(v=0, s=0, d=0, g=0, m=0, b=0)
for k, l in enumerate(lines[start + 1:end], 1):
    s = l.strip()
    if not s or s.startswith(";"):
        continue
    if s.startswith(".LBB") or s.startswith("s_barrier"):
        if lo <= first <= hi:
            flush(name, c, first)
        name = s.split(":")[0] if s.startswith(".LBB") else "--barrier--"
        first = k
        c = dict(v=0, s=0, d=0, g=0, m=0, b=0)
        continue
    if s.startswith("."):
        continue
    op = s.split()[0]
    if "mfma" in op:
        c["m"] += 1
    elif op.startswith("v_"):
        c["v"] += 1
    elif op.startswith("s_cbranch") or op.startswith("s_branch"):
        c["b"] += 1
    elif op.startswith("s_"):
        c["s"] += 1
    elif op.startswith("ds_"):
        c["d"] += 1
    elif op.startswith("global_") or op.startswith("buffer_"):
        c["g"] += 1
if lo <= first <= hi:
    flush(name, c, first)
