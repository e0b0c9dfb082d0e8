#!/bin/bash
set -o pipefail
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 || exit 1
for e in 0 -1 20; do
  echo "express_permille $e"
  python tools/ab_bench.py c4 3 express_permille=$e shard_rank=3 shard_world=8 chain=1 | tail -2 | cut -c1-220
  python tools/ab_bench.py c4 3 express_permille=$e shard_rank=3 shard_world=64 | tail -1 | cut -c1-200
  python tools/ab_bench.py c2 3 express_permille=$e | tail -1 | cut -c1-200
  python tools/ab_bench.py c4 3 express_permille=$e shard_rank=1 shard_world=2 | tail -1 | cut -c1-200
done
python tools/ab_bench.py c4 3 | tail -1 | cut -c1-200
