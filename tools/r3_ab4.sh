#!/bin/bash
# ready FIFO + collecting waves: parity, then shard / C2 / C4 timings for a few schedules
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -15 || exit 1
for ct in 0 48; do
  python tools/ab_bench.py c4 3 collect_target=$ct shard_rank=3 shard_world=8
  python tools/ab_bench.py c4 3 collect_target=$ct shard_rank=3 shard_world=8 schedule=0 chunk_spp=32 chunk_tail_min=0
  python tools/ab_bench.py c4 3 collect_target=$ct shard_rank=3 shard_world=64
  python tools/ab_bench.py c2 5 collect_target=$ct
  python tools/ab_bench.py c4 3 collect_target=$ct
done
