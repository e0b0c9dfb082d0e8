#!/bin/bash
for o in "whole=0" "whole=-1"; do
python tools/ab_bench.py c4 2 shard_rank=1 shard_world=16 census=1 $o 2>&1 | tail -3 | cut -c1-3200
done
