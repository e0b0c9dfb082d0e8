#!/bin/bash
for o in "leaf_size=2" "leaf_size=3" "leaf_size=4" "leaf_size=6" "leaf_size=8"; do
  echo "== $o"
  python tools/ab_bench.py c4 3 $o 2>&1 | tail -1 | cut -c30-140
  python tools/ab_bench.py c3 3 $o 2>&1 | tail -1 | cut -c30-140
done
