#!/bin/bash
# round 3: group walk - parity first, then timings groups = 0 / 1 on C4, its 1/8 and 1/64 shards, C2
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "group_walk or cube_image or material_coverage or cornell_image" 2>&1 | tail -15 || exit 1
for g in 0 1; do
  python tools/ab_bench.py c4 3 groups=$g shard_rank=3 shard_world=8
  python tools/ab_bench.py c4 3 groups=$g shard_rank=3 shard_world=64
  python tools/ab_bench.py c2 5 groups=$g
  python tools/ab_bench.py c4 3 groups=$g
done
