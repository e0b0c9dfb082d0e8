#!/bin/bash
for w in 512 256 128 64; do
python tools/ab_bench.py c4 3 shard_rank=5 shard_world=$w tiers=1 2>&1 | tail -2 | cut -c1-200
done
python tools/ab_bench.py c2 3 2>&1 | tail -1 | cut -c30-150
