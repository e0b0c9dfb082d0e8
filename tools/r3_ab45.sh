#!/bin/bash
for v in "" _mid _flat; do
  export PT_LIB_PATH=$PWD/owl-path-tracer_amd/libmi355pt$v.so
  echo "== $v"
  for w in 8 16 64; do
  python tools/ab_bench.py c4 3 shard_rank=1 shard_world=$w 2>&1 | tail -1 | cut -c30-180
  done
  python tools/ab_bench.py c2 3 tiers=1 2>&1 | tail -2 | cut -c1-180
  python tools/ab_bench.py c2 3 whole=1 2>&1 | tail -1 | cut -c30-180
done
