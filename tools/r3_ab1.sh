#!/bin/bash
# round 3, first A/B of the cooperative node fetch: parity tests, then C4 / 1/8 shard / C2 with coop = 0 and 1
set -o pipefail
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for c in 0 1; do
  python tools/ab_bench.py c4 3 coop=$c
  python tools/ab_bench.py c4 3 coop=$c shard_rank=3 shard_world=8
  python tools/ab_bench.py c2 5 coop=$c
done
