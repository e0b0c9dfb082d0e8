#!/bin/bash
set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cornell_image or cube_image or group_walk" 2>&1 | tail -3 || exit 1
python tools/ab_bench.py c4 5 | tail -1 | cut -c1-200
python tools/ab_bench.py c4 3 shard_rank=3 shard_world=8 | tail -1 | cut -c1-200
python tools/ab_bench.py c2 5 | tail -1 | cut -c1-200
python tools/ab_bench.py c3 3 | tail -1 | cut -c1-200
