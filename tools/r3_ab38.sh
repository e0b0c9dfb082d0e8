#!/bin/bash
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 || exit 1
for o in "whole=0" "whole=-1" "whole=0" "whole=-1"; do
  echo "== $o"
  python tools/ab_bench.py c4 3 shard_rank=1 shard_world=8 $o 2>&1 | tail -1 | cut -c30-260
  python tools/ab_bench.py c4 3 shard_rank=1 shard_world=64 $o 2>&1 | tail -1 | cut -c30-260
  python tools/ab_bench.py c2 3 $o 2>&1 | tail -1 | cut -c30-260
  python tools/ab_bench.py c4 3 $o 2>&1 | tail -1 | cut -c30-260
done
python tools/ab_bench.py c4 2 shard_rank=1 shard_world=8 latency=1 finish=1 tiers=1 2>&1 | tail -5 | cut -c1-2800
python tools/ab_bench.py c2 2 latency=1 finish=1 tiers=1 2>&1 | tail -4 | cut -c1-2800
