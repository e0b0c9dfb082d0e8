#!/bin/bash
python tools/ab_bench.py c4 2 shard_rank=1 shard_world=8 latency=1 finish=1 cost_radius=0 2>&1 | tail -3 | head -1 | cut -c1-3000
python tools/ab_bench.py c2 2 latency=1 finish=1 cost_radius=0 2>&1 | tail -2 | head -1 | cut -c1-3000
python tools/ab_bench.py c4 2 shard_rank=1 shard_world=64 latency=1 finish=1 cost_radius=0 2>&1 | tail -3 | head -1| cut -c1-3000
