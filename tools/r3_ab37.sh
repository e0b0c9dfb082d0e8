#!/bin/bash
for n in 4 8 12 16 24 32 48; do
echo "== ns_express=$n"
python tools/ab_bench.py c4 1 shard_rank=1 shard_world=8 latency=1 finish=1 whole=0 express_permille=30 ns_express=$n 2>&1 | grep -o '"us_per_ray_express[^]]*]\|"finish_ms_express[^]]*]\|"kernel_ms_min": [0-9.]*' | tr '\n' ' '; echo
done
