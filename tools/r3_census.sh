#!/bin/bash
# scheduler census of the final build (instrumented instance): C4, its 1/8 shard, C2 -> gpurun_out/r3_census.json lines
for o in "c4" "c4 shard_rank=3 shard_world=8" "c2"; do
  set -- $o
  python tools/ab_bench.py $1 2 census=1 ${@:2} | grep node_steps
done
