#!/bin/bash
# timing experiment (GPU box): edges per block of k_csr_count at a batch of 4 shards
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for per in 256 512 1024; do
  echo "== $per edges per block"
  (cd $R && GFY_EXTRA_FLAGS="-DGFY_COUNT_EDGES_PER_BLOCK=$per" python -m ginfinity_amd.build --force > /dev/null 2>&1) || exit 1
  rm -rf /tmp/cv && GFY_BENCH_STREAMS=1 timeout -k 10 120 rocprofv3 --kernel-trace --stats -d /tmp/cv -o kt --output-format csv -- $R/tools/gfy_bench 240000 100 > /dev/null 2>&1 || exit 1
  python3 - <<'PY'
import csv, glob
for path in glob.glob('/tmp/cv/**/kt_kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        if 'csr_count' in r['Name'] or 'setup_coo' in r['Name']:
            print("  %-40s avg %.1f us min %.1f" % (r['Name'][:40], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
done
