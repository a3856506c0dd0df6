#!/bin/bash
# Registers / spills of the pairwise kernels as hipcc allocates them (no GPU needed); the
# assembly is left in /tmp/pairwise.s
set -e
cd "$(dirname "$0")/../ginfinity_amd/csrc"
OUT=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math \
  -Wall -Wno-unused-function "$@" --save-temps=obj -c pairwise.hip -o "$OUT/p.o"
python3 - "$OUT/pairwise-hip-amdgcn-amd-amdhsa-gfx950.s" <<'PY'
import re, sys
text = open(sys.argv[1]).read()
meta = text[text.index('amdhsa.kernels:'):]
for block in meta.split('  - .agpr_count:')[1:]:
    name = re.search(r'\.name:\s+(\S+)', block).group(1)
    field = lambda k: re.search(r'\.%s:\s+(\d+)' % k, block).group(1)
    print(f"{name[:60]:60s} agpr {block.split()[0]:>3s} vgpr {field('vgpr_count'):>3s} "
          f"spilled {field('vgpr_spill_count'):>3s} scratch {field('private_segment_fixed_size')} B")
PY
cp "$OUT/pairwise-hip-amdgcn-amd-amdhsa-gfx950.s" /tmp/pairwise.s
rm -rf "$OUT"
