#!/bin/bash
python tools/ab_bench.py c4 2 shard_rank=1 shard_world=8 latency=1 finish=1 2>&1 | tail -3 | cut -c1-4000
python tools/ab_bench.py c2 2 latency=1 finish=1 2>&1 | tail -2 | cut -c1-4000
python tools/ab_bench.py c4 3 shard_rank=1 shard_world=64 2>&1 | tail -1 | cut -c1-300
python tools/ab_bench.py c4 3 2>&1 | tail -1 | cut -c1-300
python tools/ab_bench.py c3 3 2>&1 | tail -1 | cut -c1-300
