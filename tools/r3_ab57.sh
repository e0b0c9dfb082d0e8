#!/bin/bash
for s in 80 85 90; do for t in -1 200; do
  echo "== sticky_pct=$s chunk_tail_min=$t"
  python tools/ab_bench.py c4 3 sticky_pct=$s chunk_tail_min=$t 2>&1 | tail -1 | cut -c30-160
  python tools/ab_bench.py c3 3 sticky_pct=$s chunk_tail_min=$t 2>&1 | tail -1 | cut -c30-160
done; done
for s in -1 30 50 65 80; do
  echo "== shards sticky_pct=$s"
  python tools/ab_bench.py c4 2 shard_rank=1 shard_world=2 sticky_pct=$s 2>&1 | tail -1 | cut -c30-160
  python tools/ab_bench.py c4 2 shard_rank=1 shard_world=4 sticky_pct=$s 2>&1 | tail -1 | cut -c30-160
done
