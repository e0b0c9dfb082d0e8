#!/bin/bash
for o in "" "express_permille=5 ns_express=48" "express_permille=10 ns_express=48" "express_permille=20 ns_express=64" "express_permille=10 ns_express=32" "express_permille=30 ns_express=64"; do
  echo "== $o"
  python tools/ab_bench.py c4 3 $o 2>&1 | tail -1 | cut -c30-160
done
