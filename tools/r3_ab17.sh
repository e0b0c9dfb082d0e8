#!/bin/bash
for nse in 2 4 8 16; do
  echo "ns_express $nse"
  python tools/ab_bench.py c4 3 ns_express=$nse shard_rank=3 shard_world=8 chain=1 | tail -2 | cut -c1-220
  python tools/ab_bench.py c4 3 ns_express=$nse shard_rank=3 shard_world=64 | tail -1 | cut -c1-200
  python tools/ab_bench.py c2 3 ns_express=$nse | tail -1 | cut -c1-200
done
