#!/bin/bash
python tools/ab_bench.py c4 3 2>&1 | tail -1 | cut -c30-130
python tools/ab_bench.py c3 3 2>&1 | tail -1 | cut -c30-130
python tools/ab_bench.py c2 3 2>&1 | tail -1 | cut -c30-130
python tools/ab_bench.py c4 2 shard_rank=1 shard_world=2 2>&1 | tail -1 | cut -c30-160
python tools/ab_bench.py c4 2 shard_rank=1 shard_world=4 2>&1 | tail -1 | cut -c30-160
python tools/ab_bench.py c4 2 shard_rank=1 shard_world=8 2>&1 | tail -1 | cut -c30-160
python -m pytest tests/test_gpu_parity.py -q -s -m gpu -k "c5_car or c1" 2>&1 | grep "kernel_ms\|passed"
