#!/bin/bash
out=gpurun_out/r2_sweep4.log
: > $out
V=owl-path-tracer_amd/variants
for lib in $V/lib_base.so $V/lib_nopf.so owl-path-tracer_amd/libmi355pt.so; do
  o="fat=0"; [ $lib = $V/lib_base.so ] && o=""
  PT_LIB_PATH=$lib python tools/ab_bench.py c4 2 $o 2>&1 | tail -1 >> $out
  PT_LIB_PATH=$lib python tools/ab_bench.py c4 2 shard_rank=5 shard_world=8 $o 2>&1 | tail -1 >> $out
  PT_LIB_PATH=$lib python tools/ab_bench.py c2 3 $o 2>&1 | tail -1 >> $out
done
for o in "slots_per_wave=32 blocks_per_cu=8" "slots_per_wave=32 blocks_per_cu=16" "slots_per_wave=48 blocks_per_cu=8" "slots_per_wave=48 blocks_per_cu=12" "slots_per_wave=48 blocks_per_cu=16" "slots_per_wave=32 blocks_per_cu=12 tune1=8" "slots_per_wave=48 blocks_per_cu=12 tune1=12 tune2=16" "slots_per_wave=24 blocks_per_cu=16" "slots_per_wave=32 blocks_per_cu=12 sticky_pct=50"; do
  python tools/ab_bench.py c4 2 shard_rank=5 shard_world=8 fat=0 $o 2>&1 | tail -1 >> $out
done
for o in "slots_per_wave=32 blocks_per_cu=16" "slots_per_wave=48 blocks_per_cu=12" "slots_per_wave=48 blocks_per_cu=16" "slots_per_wave=64 blocks_per_cu=12"; do
  python tools/ab_bench.py c2 3 fat=0 $o 2>&1 | tail -1 >> $out
done
for o in "leaf_size=2" "leaf_size=3" "leaf_size=6"; do
  python tools/ab_bench.py c4 2 fat=0 $o 2>&1 | tail -1 >> $out
done
