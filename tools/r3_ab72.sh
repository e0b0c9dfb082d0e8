#!/bin/bash
for o in "" "express_permille=0" "express_permille=10" "express_permille=20" "express_permille=20 ns_express=4" "express_permille=40" ; do
  echo "== $o"
  python tools/ab_bench.py c4 2 shard_rank=1 shard_world=2 $o 2>&1 | tail -1 | cut -c30-190
  python tools/ab_bench.py c4 2 shard_rank=1 shard_world=4 $o 2>&1 | tail -1 | cut -c30-190
done
