#!/bin/bash
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5 || exit 1
for q in 0 1; do
  python tools/ab_bench.py c4 3 quant=$q census=1 | grep -v node_steps | cut -c1-250
  python tools/ab_bench.py c4 3 quant=$q shard_rank=3 shard_world=8 | tail -1 | cut -c1-200
  python tools/ab_bench.py c2 3 quant=$q | tail -1 | cut -c1-200
done
python tools/ab_bench.py c3 3 quant=0 | tail -1 | cut -c1-200
python tools/ab_bench.py c3 3 quant=1 | tail -1 | cut -c1-200
