#!/bin/bash
# fat (two-level) steps vs plain steps, both with the register-cached stack top and whole-leaf prefetch
out=gpurun_out/r2_sweep3.log
: > $out
python -m pytest tests -x -q -m gpu -k "closest_hit or cube_image or cornell_image or material_coverage or chunked" > gpurun_out/r2_sweep3_tests.log 2>&1 || { tail -30 gpurun_out/r2_sweep3_tests.log; exit 1; }
tail -2 gpurun_out/r2_sweep3_tests.log
for o in "fat=1" "fat=0" "fat=1 tune0=16" "fat=1 tune0=32"; do
  python tools/ab_bench.py c4 2 $o 2>&1 | tail -1 >> $out
  python tools/ab_bench.py c4 2 shard_rank=5 shard_world=8 $o 2>&1 | tail -1 >> $out
  python tools/ab_bench.py c2 3 $o 2>&1 | tail -1 >> $out
done
python tools/ab_bench.py c4 1 fat=1 census=1 2>&1 | tail -2 | head -1 >> $out
