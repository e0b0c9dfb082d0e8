#!/bin/bash
out=gpurun_out/r2_sweep10.log
: > $out
for sh in "shard_rank=5 shard_world=8" "shard_rank=5 shard_world=64"; do
for o in "" "tune5=1" "tune5=1 tune6=64" "tune5=1 tune6=80" "tune5=1 tune6=64 schedule=0 chunk_spp=32"; do
  python tools/ab_bench.py c4 2 $sh $o 2>&1 | tail -1 >> $out
done; done
for o in "" "tune5=1" "tune5=1 tune6=64"; do
  python tools/ab_bench.py c4 2 $o 2>&1 | tail -1 >> $out
  python tools/ab_bench.py c2 3 $o 2>&1 | tail -1 >> $out
done
python tools/ab_bench.py c4 2 shard_rank=5 shard_world=64 tune5=1 census=1 2>&1 | tail -3 | head -2 | cut -c1-2500 >> $out
