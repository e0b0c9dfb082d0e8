#!/bin/bash
set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cornell_image or cube_image" 2>&1 | tail -3 || exit 1
for t in 0 1; do
  python tools/ab_bench.py c4 3 tune4=$t | tail -1 | cut -c1-200
  python tools/ab_bench.py c4 3 tune4=$t shard_rank=3 shard_world=8 | tail -1 | cut -c1-200
  python tools/ab_bench.py c2 3 tune4=$t | tail -1 | cut -c1-200
done
