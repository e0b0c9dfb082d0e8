#!/bin/bash
for o in "" "slots_per_wave=64" "slots_per_wave=72" "slots_per_wave=80" "slots_per_wave=104" "whole=0" "whole=0 slots_per_wave=64" "whole=0 slots_per_wave=80"; do
  echo "== $o"
  python tools/ab_bench.py c2 3 $o 2>&1 | tail -1 | cut -c30-200
done
