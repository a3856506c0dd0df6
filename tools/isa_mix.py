"""Static instruction mix of one kernel in a hipcc -S listing, split at s_barrier.

    python tools/isa_mix.py layer.s k_gine_layer_f16ILb1ELb0

Counts are per static segment (program order), so loop tails / cold blocks laid out
after a barrier are attributed to the segment they are printed in.
"""
import collections
import re
import sys

path, needle = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and needle in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
seg, segs = collections.Counter(), []
labels = []
for l in lines[start + 1:end]:
    s = l.strip()
    if not s or s.startswith(";") or s.startswith("."):
        if s.startswith(".LBB"):
            labels.append(s.split(":")[0])
        continue
    op = s.split()[0]
    if op == "s_barrier":
        segs.append((seg, labels))
        seg, labels = collections.Counter(), []
        continue
    seg[op] += 1
segs.append((seg, labels))
for i, (c, lab) in enumerate(segs):
    total = sum(c.values())
    valu = sum(v for k, v in c.items() if k.startswith("v_") and "mfma" not in k)
    salu = sum(v for k, v in c.items() if k.startswith("s_"))
    lds = sum(v for k, v in c.items() if k.startswith("ds_"))
    vmem = sum(v for k, v in c.items() if k.startswith("global_") or k.startswith("buffer_"))
    mfma = sum(v for k, v in c.items() if "mfma" in k)
    print(f"seg {i}: {total} instr  valu {valu} salu {salu} lds {lds} vmem {vmem} mfma {mfma}"
          f"  blocks {lab[:1]}..{lab[-1:]} ({len(lab)})")
    if "-v" in sys.argv:
        print("   ", ", ".join(f"{k} {v}" for k, v in c.most_common(14)))
