#!/bin/bash
for o in "xcd_block=0" "xcd_block=8" "xcd_block=10" "xcd_block=12" "xcd_block=14" "xcd_block=0"; do
  echo "== $o"
  python tools/ab_bench.py c4 3 $o 2>&1 | tail -1 | cut -c30-140
  python tools/ab_bench.py c3 3 $o 2>&1 | tail -1 | cut -c30-140
  python tools/ab_bench.py c4 2 shard_rank=1 shard_world=2 $o 2>&1 | tail -1 | cut -c30-170
done
