#!/bin/bash
for o in "" "express_permille=30" "express_permille=80" "express_permille=80 ns_express=16" "express_permille=80 ns_express=32" "express_permille=150 ns_express=32" "express_permille=150 ns_express=48" "whole=1"; do
  echo "== $o"
  python tools/ab_bench.py c2 3 $o 2>&1 | tail -1 | cut -c30-200
done
