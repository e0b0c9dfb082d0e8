#!/bin/bash
for v in "" _r16 _r32; do
  export PT_LIB_PATH=$PWD/owl-path-tracer_amd/libmi355pt$v.so
  echo "== $v"
  python tools/ab_bench.py c4 2 shard_rank=1 shard_world=8 2>&1 | tail -1 | cut -c30-160
  python tools/ab_bench.py c4 2 shard_rank=1 shard_world=16 2>&1 | tail -1 | cut -c30-160
  python tools/ab_bench.py c4 2 shard_rank=1 shard_world=64 2>&1 | tail -1 | cut -c30-160
  python tools/ab_bench.py c2 3 2>&1 | tail -1 | cut -c30-130
  python tools/ab_bench.py c4 3 2>&1 | tail -1 | cut -c30-130
done
