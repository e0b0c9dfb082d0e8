#!/bin/bash
python tools/ab_bench.py c4 2 shard_rank=1 shard_world=8 latency=1 finish=1 2>&1 | tail -3 | head -1 | cut -c1-4200
