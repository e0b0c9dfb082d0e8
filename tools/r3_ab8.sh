#!/bin/bash
set -o pipefail
python tools/ab_bench.py c4 2 shard_rank=3 shard_world=8 census=1 | grep node_steps | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['groups'], d['cycle_share'])
"
for t in "8 16" "16 24" "24 32" "32 48" "48 64" "64 96"; do
  set -- $t
  echo "max_rays $1 max_run $2"
  python tools/ab_bench.py c4 3 shard_rank=3 shard_world=8 tune6=$1 tune7=$2 | tail -1 | cut -c1-200
  python tools/ab_bench.py c2 3 tune6=$1 tune7=$2 | tail -1 | cut -c1-160
done
