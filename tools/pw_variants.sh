#!/bin/bash
# timing experiments on k_pairwise (GPU box): rebuild with a flag, run the distance bench
for flags in "-DGFY_PAIRWISE_REQUEST_AFTER_MULTIPLY=2" "-DGFY_PAIRWISE_REQUEST_AFTER_MULTIPLY=1" "-DGFY_PAIRWISE_REQUEST_AFTER_MULTIPLY=0"; do
  echo "== flags: $flags"
  GFY_EXTRA_FLAGS="$flags" python -m ginfinity_amd.build --force > /dev/null 2>&1 || exit 1
  timeout -k 10 120 python tools/bench_distance.py --repeats 2 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['seconds'], d['roofline']['frac'])" || exit 1
done
