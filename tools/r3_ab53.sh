#!/bin/bash
for o in "" "chunk_tail_min=8" "chunk_tail_min=32" "chunk_tail_min=4" "sticky_pct=70" "sticky_pct=80" "prepass_spp=6" ""; do
  echo "== $o"
  python tools/ab_bench.py c4 3 $o 2>&1 | tail -1 | cut -c30-120
done
