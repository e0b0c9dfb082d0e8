#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "express or shard or group" 2>&1 | tail -3 || exit 1
for o in "express_cus=0" "express_cus=-1" "express_cus=2" "express_cus=4" "express_cus=3 ns_express=4" "express_cus=0"; do
  echo "== $o"
  python tools/ab_bench.py c4 3 shard_rank=1 shard_world=8 $o | tail -1 | cut -c30-200
  python tools/ab_bench.py c4 3 shard_rank=1 shard_world=64 $o | tail -1 | cut -c30-200
  python tools/ab_bench.py c2 3 $o | tail -1 | cut -c30-200
done
