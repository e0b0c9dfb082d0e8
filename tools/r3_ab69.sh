#!/bin/bash
for e in "" "1" "" "1"; do
  echo "== PT_DBG_NS104=$e"
  if [ -n "$e" ]; then export PT_DBG_NS104=$e; else unset PT_DBG_NS104; fi
  python tools/ab_bench.py c2 3 2>&1 | tail -1 | cut -c30-190
  for w in 8 16 64; do
  python tools/ab_bench.py c4 2 shard_rank=1 shard_world=$w 2>&1 | tail -1 | cut -c30-210
  done
done
