#!/bin/bash
for o in "sticky_pct=10" "sticky_pct=20" "sticky_pct=50" "sticky_pct=75" "sticky_pct=95"; do
  echo "== $o"
  python tools/ab_bench.py c4 3 shard_rank=1 shard_world=8 $o 2>&1 | tail -1 | cut -c30-200
done
python tools/ab_bench.py c4 2 shard_rank=1 shard_world=8 latency=1 finish=1 sticky_pct=95 2>&1 | tail -3 | cut -c1-2600
