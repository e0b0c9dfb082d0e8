#!/bin/bash
for f in 8 6 5; do
  echo "== rest/$f"
  python tools/ab_bench.py c4 3 chunk_tail_min=$((1016/f)) 2>&1 | tail -1 | cut -c30-130
  python tools/ab_bench.py c3 3 chunk_tail_min=$((504/f)) 2>&1 | tail -1 | cut -c30-130
  python tools/ab_bench.py c2 3 chunk_tail_min=$((240/f)) 2>&1 | tail -1 | cut -c30-130
  python tools/ab_bench.py c4 2 shard_rank=1 shard_world=2 chunk_tail_min=$((1016/f)) 2>&1 | tail -1 | cut -c30-160
  python tools/ab_bench.py c4 2 shard_rank=1 shard_world=4 chunk_tail_min=$((1016/f)) 2>&1 | tail -1 | cut -c30-160
done
