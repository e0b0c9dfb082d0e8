#!/bin/bash
for s in 13 30 45 60; do
  echo "== c2 sticky_pct=$s"
  python tools/ab_bench.py c2 3 sticky_pct=$s 2>&1 | tail -1 | cut -c30-130
done
for s in 53 60 67 75; do
  echo "== c3 sticky_pct=$s"
  python tools/ab_bench.py c3 3 sticky_pct=$s 2>&1 | tail -1 | cut -c30-130
done
echo "== c5"
python -m pytest tests/test_gpu_parity.py -q -s -m gpu -k "c5_car" 2>&1 | grep "kernel_ms"
