#!/bin/bash
python tools/ab_bench.py c4 2 shard_rank=1 shard_world=8 latency=1 finish=1 express_permille=40 2>&1 | tail -2 | cut -c1-2500
for o in "express_permille=10" "express_permille=20" "express_permille=40" "express_permille=80" "express_permille=40 ns_express=16" "express_permille=80 ns_express=16" "express_permille=160 ns_express=24"  "express_permille=160 ns_express=32"; do
  echo "== $o"
  python tools/ab_bench.py c4 3 shard_rank=1 shard_world=8 $o 2>&1 | tail -1 | cut -c30-300
done
