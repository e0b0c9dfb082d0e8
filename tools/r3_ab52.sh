#!/bin/bash
for o in "" "express_permille=1" "express_permille=2" "express_permille=5" "express_permille=10" "express_permille=2 ns_express=4" "express_permille=5 ns_express=16"; do
  echo "== $o"
  python tools/ab_bench.py c4 3 $o 2>&1 | tail -1 | cut -c30-200
done
python tools/ab_bench.py c4 1 latency=1 finish=1 express_permille=2 2>&1 | tail -3 | head -2 | cut -c1-2500
