#!/bin/bash
for o in "whole=0" "whole=-1" "whole=0" "whole=-1"; do
  echo "== $o"
  python tools/ab_bench.py c2 5 $o tiers=1 2>&1 | tail -2 | cut -c1-260
done
