#!/bin/bash
set -o pipefail
for ct in 0 16 96; do
  python tools/ab_bench.py c4 2 collect_target=$ct shard_rank=3 shard_world=8 census=1 chain=1 | grep -v '^{"rays"' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if 'node_steps' in d:
        print({k: d[k] for k in ('node_lanes/node_steps','tri_lanes/tri_steps','hit_items/hit_passes','miss_items/miss_passes','idle_sum/iters','active_sum/iters','wait_polls','sleeps','cycle_share','groups','winddown_time_share')})
    else:
        print(d)
"
done
