#!/bin/bash
for o in "" "slots_per_wave=64" "slots_per_wave=72" "slots_per_wave=80"; do
  echo "== $o"
  python tools/ab_bench.py c4 3 shard_rank=1 shard_world=8 $o 2>&1 | tail -1 | cut -c30-400
  python tools/ab_bench.py c4 3 shard_rank=1 shard_world=4 $o 2>&1 | tail -1 | cut -c30-400
  python tools/ab_bench.py c2 3 $o 2>&1| tail -1 | cut -c30-400
done
