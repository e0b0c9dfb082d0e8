#!/bin/bash
for o in "whole=0" "whole=-1"; do
  echo "== $o"
  for w in 8 16 32 64; do
  python tools/ab_bench.py c4 3 shard_rank=1 shard_world=$w $o 2>&1 | tail -1 | cut -c30-180
  done
  python tools/ab_bench.py c4 3 shard_rank=1 shard_world=4 $o 2>&1 | tail -1 | cut -c30-180
  python tools/ab_bench.py c2 3 $o 2>&1 | tail -1 | cut -c30-180
done
