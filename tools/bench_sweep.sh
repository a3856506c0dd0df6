#!/bin/bash
# bench.py over batch sizes and batches in flight (GPU box): nodes/s per configuration
R=${GRAFT_REPO_ROOT:-/root/repo}
for B in 2 3 4 5 6 8; do for S in 1 2 3; do
  timeout -k 10 120 python3 $R/bench.py --steps 960 --warmup 96 --batch $B --streams $S --no-cpu-baseline --distance-rows 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('batch $B streams $S: %.1f M nodes/s  %.2f us/shard  layer isolated %.3f' % (d['value']/1e6, d['ms_per_step']*1e3, d['roofline']['isolated']['frac']))" || exit 1
done; done
