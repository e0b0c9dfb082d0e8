#!/bin/bash
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 || exit 1
for v in "" _prev "" _prev; do
  export PT_LIB_PATH=$PWD/owl-path-tracer_amd/libmi355pt$v.so
  echo "== $v"
  python tools/ab_bench.py c4 3 | tail -1 | cut -c30-140
  python tools/ab_bench.py c4 3 shard_rank=1 shard_world=8 | tail -1 | cut -c30-140
  python tools/ab_bench.py c4 3 shard_rank=1 shard_world=64 | tail -1 | cut -c30-140
  python tools/ab_bench.py c2 3 | tail -1 | cut -c30-140
done
