#!/bin/bash
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 || exit 1
python tools/ab_bench.py c4 3 2>&1 | tail -1 | cut -c30-130
python tools/ab_bench.py c3 3 2>&1 | tail -1 | cut -c30-130
python tools/ab_bench.py c2 3 2>&1 | tail -1 | cut -c30-130
python tools/ab_bench.py c4 2 shard_rank=1 shard_world=2 2>&1 | tail -1 | cut -c30-160
python tools/ab_bench.py c4 2 shard_rank=1 shard_world=4 2>&1 | tail -1 | cut -c30-160
python tools/ab_bench.py c4 2 shard_rank=1 shard_world=8 2>&1 | tail -1 | cut -c30-160
python tools/ab_bench.py c4 2 shard_rank=1 shard_world=64 2>&1 | tail -1 | cut -c30-160
python tools/ab_bench.py c4 1 latency=1 finish=1 2>&1 | tail -3 | head -2 | cut -c1-1500
