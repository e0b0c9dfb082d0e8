#!/bin/bash
for o in "" "tune6=8 tune7=16" "tune6=12 tune7=16" "tune6=24 tune7=32" "tune6=32 tune7=48"; do
  echo "== $o"
  python tools/ab_bench.py c4 3 shard_rank=1 shard_world=8 $o 2>&1 | tail -1 | cut -c30-170
  python tools/ab_bench.py c4 2 $o 2>&1 | tail -1 | cut -c30-140
done
