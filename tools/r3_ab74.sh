#!/bin/bash
for pm in 0 10 30 100 300; do
  echo "== prio_permille=$pm"
  python tools/ab_bench.py c4 3 prio_permille=$pm 2>&1 | tail -1 | cut -c30-150
  python tools/ab_bench.py c3 3 prio_permille=$pm 2>&1 | tail -1 | cut -c30-150
  python tools/ab_bench.py c4 2 shard_rank=1 shard_world=2 prio_permille=$pm 2>&1 | tail -1 | cut -c30-180
  python tools/ab_bench.py c4 2 shard_rank=1 shard_world=4 prio_permille=$pm 2>&1 | tail -1 | cut -c30-180
  python tools/ab_bench.py c2 3 prio_permille=$pm 2>&1 | tail -1 | cut -c30-150
done
