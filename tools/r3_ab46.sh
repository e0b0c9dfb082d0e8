#!/bin/bash
for v in "" _a _b _c; do
  export PT_LIB_PATH=$PWD/owl-path-tracer_amd/libmi355pt$v.so
  echo "== $v"
  for w in 8 16 32; do
  python tools/ab_bench.py c4 3 shard_rank=1 shard_world=$w 2>&1 | tail -1 | cut -c30-180
  done
done
