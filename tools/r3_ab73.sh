#!/bin/bash
for o in "whole=0" "whole=-1"; do
  echo "== $o"
  python tools/ab_bench.py c3 3 shard_rank=1 shard_world=8 $o tiers=1 2>&1 | tail -1 | cut -c30-170
  python tools/ab_bench.py c3 3 shard_rank=1 shard_world=4 $o 2>&1 | tail -1 | cut -c30-170
  python tools/ab_bench.py c3 3 shard_rank=1 shard_world=64 $o 2>&1 | tail -1 | cut -c30-170
  python tools/ab_bench.py c5 2 shard_rank=1 shard_world=8 spp=1024 $o 2>&1 | tail -1 | cut -c30-190
  python tools/ab_bench.py c5 2 shard_rank=1 shard_world=64 spp=1024 $o 2>&1 | tail -1 | cut -c30-190
  python tools/ab_bench.py c2 3 shard_rank=1 shard_world=8 $o 2>&1 | tail -1 | cut -c30-170
done
