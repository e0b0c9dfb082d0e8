#!/bin/bash
for t in 16 24 32 48 64 96 128 0; do
  echo "== chunk_tail_min=$t"
  python tools/ab_bench.py c4 3 chunk_tail_min=$t 2>&1 | tail -1 | cut -c30-120
  python tools/ab_bench.py c3 3 chunk_tail_min=$t 2>&1 | tail -1 | cut -c30-120
  python tools/ab_bench.py c4 2 shard_rank=1 shard_world=4 chunk_tail_min=$t 2>&1 | tail -1 | cut -c30-150
  python tools/ab_bench.py c2 3 chunk_tail_min=$t 2>&1 | tail -1 | cut -c30-120
done
