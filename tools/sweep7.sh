#!/bin/bash
out=gpurun_out/r2_sweep7.log
: > $out
python -m pytest tests -x -q -m gpu -k "closest_hit or cube_image or cornell_image or material_coverage" > gpurun_out/r2_sweep7_tests.log 2>&1 || { tail -30 gpurun_out/r2_sweep7_tests.log; exit 1; }
tail -1 gpurun_out/r2_sweep7_tests.log
V=owl-path-tracer_amd/variants
for o in "" "leaf_align=4" "node_pairs=1" "node_pairs=1 leaf_align=4" "leaf_align=8"; do
  python tools/ab_bench.py c4 2 $o 2>&1 | tail -1 >> $out
done
PT_LIB_PATH=$V/lib_base.so python tools/ab_bench.py c4 2 2>&1 | tail -1 >> $out
for o in "" "node_pairs=1 leaf_align=4"; do
  python tools/ab_bench.py c4 2 shard_rank=5 shard_world=8 $o 2>&1 | tail -1 >> $out
  python tools/ab_bench.py c2 3 $o 2>&1 | tail -1 >> $out
done
PT_LIB_PATH=$V/lib_base.so python tools/ab_bench.py c4 2 shard_rank=5 shard_world=8 2>&1 | tail -1 >> $out
PT_LIB_PATH=$V/lib_base.so python tools/ab_bench.py c2 3 2>&1 | tail -1 >> $out
# near-idle GPU: how fast can a chain go?  tiny shards of C4
for o in "shard_world=512" "shard_world=2048" "shard_world=2048 slots_per_wave=32" "shard_world=2048 slots_per_wave=16" "shard_world=2048 slots_per_wave=16 tune1=4 tune2=8"; do
  python tools/ab_bench.py c4 2 shard_rank=5 $o 2>&1 | tail -2 | cut -c1-400 >> $out
done
