#!/bin/bash
for o in "cost_radius=0" "cost_radius=1"; do
  echo "== $o"
  python tools/ab_bench.py c4 2 shard_rank=1 shard_world=8 latency=1 finish=1 tiers=1 $o 2>&1 | tail -5 | cut -c1-2300
  python tools/ab_bench.py c2 2 tiers=1 $o 2>&1 | tail -2 | cut -c1-1500
done
