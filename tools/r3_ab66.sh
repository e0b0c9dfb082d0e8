#!/bin/bash
for e in "" "1" "2"; do
  echo "== PT_DBG_RING_GRID=$e"
  if [ -n "$e" ]; then export PT_DBG_RING_GRID=$e; fi
  python tools/ab_bench.py c2 5 2>&1 | tail -1 | cut -c30-180
done
unset PT_DBG_RING_GRID
python tools/ab_bench.py c2 5 whole=0 2>&1 | tail -1 | cut -c30-180
