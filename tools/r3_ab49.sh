#!/bin/bash
for o in "" "prepass_spp=8" "prepass_spp=24" "prepass_spp=32"; do
  echo "== $o"
  for w in 8 16 64; do
  python tools/ab_bench.py c4 3 shard_rank=1 shard_world=$w $o 2>&1 | tail -1 | cut -c30-160
  done
  python tools/ab_bench.py c2 3 $o 2>&1 | tail -1 | cut -c30-160
done
