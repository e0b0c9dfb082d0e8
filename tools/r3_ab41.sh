#!/bin/bash
for o in "cost_radius=2" "cost_radius=3"; do
python tools/ab_bench.py c4 2 shard_rank=1 shard_world=8 latency=1 finish=1 $o 2>&1 | tail -3 | cut -c1-3000
done
python tools/ab_bench.py c2 2 latency=1 finish=1 2>&1 | tail -2 | cut -c1-3000
python tools/ab_bench.py c4 3 2>&1 | tail -1 | cut -c1-300
