#!/bin/bash
out=gpurun_out/r2_sweep8.log
: > $out
for o in "" "leaf_align=4" "node_pairs=1" "node_pairs=1 leaf_align=4" "leaf_align=8"; do
  python tools/ab_bench.py c4 2 $o 2>&1 | tail -1 >> $out
done
python tools/ab_bench.py c2 3 node_pairs=1 leaf_align=4 2>&1 | tail -1 >> $out
python tools/ab_bench.py c4 2 shard_rank=5 shard_world=8 node_pairs=1 leaf_align=4 2>&1 | tail -1 >> $out
for o in "" "shard_rank=5 shard_world=8" "shard_rank=5 shard_world=64" "shard_rank=5 shard_world=512"; do
  python tools/ab_bench.py c4 2 chain=1 census=1 $o 2>&1 | tail -4 | cut -c1-3000 >> $out
done
python tools/ab_bench.py c2 3 chain=1 2>&1 | tail -2 >> $out
