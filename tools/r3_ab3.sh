#!/bin/bash
# group walk census
set -o pipefail
python tools/ab_bench.py c4 2 groups=1 shard_rank=3 shard_world=8 census=1 chain=1
python tools/ab_bench.py c4 2 groups=1 shard_rank=3 shard_world=64 census=1 chain=1
python tools/ab_bench.py c2 3 groups=1 census=1 chain=1
