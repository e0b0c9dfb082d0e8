#!/bin/bash
export PT_DEBUG_EXPRESS=1
for o in "express_cus=0" "express_cus=-1" "express_cus=2" "express_cus=4"; do
  echo "== $o"
  python tools/ab_bench.py c4 1 shard_rank=1 shard_world=8 $o 2>&1 | tail -4 | cut -c1-400
  python tools/ab_bench.py c4 1 shard_rank=1 shard_world=64 $o 2>&1| tail -4 | cut -c1-400
done
