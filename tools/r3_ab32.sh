#!/bin/bash
for o in "express_permille=10" "express_permille=40" "express_permille=80"; do
python tools/ab_bench.py c4 2 shard_rank=1 shard_world=8 latency=1 finish=1 $o 2>&1 | tail -3 | cut -c1-2600
done
python tools/ab_bench.py c2 2 latency=1 finish=1 express_permille=40 2>&1 | tail -2 | cut -c1-2600
