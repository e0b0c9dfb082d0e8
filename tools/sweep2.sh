#!/bin/bash
# zero-code sweep: fewer, denser waves in the latency regime (blocks_per_cu x slots_per_wave)
out=gpurun_out/r2_sweep2.log
: > $out
for o in "blocks_per_cu=4" "blocks_per_cu=8" "blocks_per_cu=12" "blocks_per_cu=4 slots_per_wave=128" "blocks_per_cu=8 slots_per_wave=128" "blocks_per_cu=8 slots_per_wave=160" "blocks_per_cu=4 slots_per_wave=192" "blocks_per_cu=8 slots_per_wave=128 sticky_pct=50" "blocks_per_cu=8 slots_per_wave=128 sticky_pct=75"; do
  python tools/ab_bench.py c4 2 shard_rank=5 shard_world=8 $o 2>&1 | tail -1 >> $out
done
for o in "blocks_per_cu=4" "blocks_per_cu=8" "blocks_per_cu=4 slots_per_wave=128" "blocks_per_cu=8 slots_per_wave=128" "blocks_per_cu=4 slots_per_wave=192" "blocks_per_cu=8 slots_per_wave=128 sticky_pct=50" "blocks_per_cu=4 slots_per_wave=255 sticky_pct=50"; do
  python tools/ab_bench.py c2 3 $o 2>&1 | tail -1 >> $out
done
for o in "blocks_per_cu=12 slots_per_wave=128" "blocks_per_cu=10 slots_per_wave=160"; do
  python tools/ab_bench.py c4 2 $o 2>&1 | tail -1 >> $out
done
