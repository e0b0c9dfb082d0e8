#!/bin/bash
for lib in libmi355pt.so libmi355pt_unord.so; do
  export PT_LIB_PATH=$PWD/owl-path-tracer_amd/$lib
  echo "== $lib"
  python tools/ab_bench.py c4 3 shard_rank=3 shard_world=8 | tail -1 | cut -c1-200
  python tools/ab_bench.py c4 3 shard_rank=3 shard_world=64 | tail -1 | cut -c1-200
  python tools/ab_bench.py c2 3 | tail -1 | cut -c1-200
  python tools/ab_bench.py c4 3 | tail -1 | cut -c1-200
done
