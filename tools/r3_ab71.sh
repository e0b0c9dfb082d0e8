#!/bin/bash
for o in "" "cost_radius=1" "cost_radius=3" "cost_radius=4" ""; do
  echo "== $o"
  for w in 8 16 64; do
  python tools/ab_bench.py c4 3 shard_rank=1 shard_world=$w $o 2>&1 | tail -1 | cut -c30-150
  done
done
