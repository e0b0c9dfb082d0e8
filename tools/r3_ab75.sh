#!/bin/bash
for o in "" "tune1=24" "tune1=40" "tune1=48" "tune2=16" "tune2=24" "tune2=48" "tune1=40 tune2=24" "tune1=24 tune2=48"; do
  echo "== $o"
  python tools/ab_bench.py c2 3 $o 2>&1 | tail -1 | cut -c30-130
  python tools/ab_bench.py c4 2 $o 2>&1 | tail -1 | cut -c30-130
done
