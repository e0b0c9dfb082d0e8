#!/bin/bash
for o in "" "sticky_pct=60" "sticky_pct=85" "chunk_spp=32" "chunk_spp=128" "cost_radius=1" "cost_radius=3" "prepass_spp=4" "prepass_spp=16" ""; do
  echo "== $o"
  python tools/ab_bench.py c4 3 $o 2>&1 | tail -1 | cut -c30-160
done
for o in "" "cost_radius=1" "cost_radius=3" "prepass_spp=4" "prepass_spp=16"; do
  echo "== c2 $o"
  python tools/ab_bench.py c2 3 $o 2>&1 | tail -1 | cut -c30-160
  python tools/ab_bench.py c4 3 shard_rank=1 shard_world=8 $o 2>&1 | tail -1 | cut -c30-160
done
