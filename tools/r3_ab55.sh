#!/bin/bash
for s in 50 65 75 85; do for t in 96 128 192 256; do
  echo "== sticky_pct=$s chunk_tail_min=$t"
  python tools/ab_bench.py c4 3 sticky_pct=$s chunk_tail_min=$t 2>&1 | tail -1 | cut -c30-160
  python tools/ab_bench.py c3 3 sticky_pct=$s chunk_tail_min=$t 2>&1 | tail -1 | cut -c30-160
done; done
