#!/bin/bash
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "group_walk or cube_image or material_coverage or cornell_image or lbvh" 2>&1 | tail -5 || exit 1
python tools/r3_visits.py c4 8
python tools/r3_visits.py c2 32
python tools/ab_bench.py c4 3 shard_rank=3 shard_world=8 census=1 chain=1 | grep -v '^{"rays"\|node_steps'
python tools/ab_bench.py c4 3 shard_rank=3 shard_world=64
python tools/ab_bench.py c2 5
python tools/ab_bench.py c4 3
