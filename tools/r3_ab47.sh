#!/bin/bash
for o in "whole=-1" "whole=1"; do
  echo "== $o"
  python tools/ab_bench.py c4 3 shard_rank=1 shard_world=8 $o 2>&1 | tail -1 | cut -c30-180
  python tools/ab_bench.py c4 3 shard_rank=1 shard_world=4 $o tiers=1 2>&1 | tail -2 | cut -c1-900
  python tools/ab_bench.py c4 3 shard_rank=1 shard_world=2 $o 2>&1 | tail -1 | cut -c30-180
  python tools/ab_bench.py c3 3 $o 2>&1 | tail -1 | cut -c30-180
  python tools/ab_bench.py c4 3 $o tiers=1 2>&1 | tail -2 | cut -c1-900
done
