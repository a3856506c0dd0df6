#!/bin/bash
# timing experiment (GPU box): occupancy hint of k_encode_setup_coo at a batch of 4 shards
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for b in 1 2 4 8; do
  echo "== min blocks $b"
  (cd $R && GFY_EXTRA_FLAGS="-DGFY_SETUP_MIN_BLOCKS=$b" python -m ginfinity_amd.build --force > /dev/null 2>&1) || exit 1
  for N in 240000 60000; do
  rm -rf /tmp/cv && GFY_BENCH_STREAMS=1 timeout -k 10 120 rocprofv3 --kernel-trace --stats -d /tmp/cv -o kt --output-format csv -- $R/tools/gfy_bench $N 100 > /dev/null 2>&1 || exit 1
  python3 - $N <<'PY'
import csv, glob, sys
for path in glob.glob('/tmp/cv/**/kt_kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        if 'setup_coo' in r['Name']:
            print("  N=%s %-40s avg %.1f us min %.1f" % (sys.argv[1], r['Name'][:40], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
  done
done
