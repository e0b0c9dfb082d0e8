#!/bin/bash
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lbvh" -s 2>&1 | tail -8 || exit 1
for b in 0 1 2; do
  python tools/ab_bench.py c4 3 bvh_builder=$b | tail -1
done
for r in 8 32; do
  python tools/ab_bench.py c4 3 bvh_builder=2 ploc_radius=$r | tail -1
done
python tools/ab_bench.py c5 2 bvh_builder=2 spp=512 | tail -1
python tools/ab_bench.py c5 2 bvh_builder=0 spp=512 | tail -1
