#!/bin/bash
# zero-code option sweep: slots per wave / sticky share in the latency regime
out=gpurun_out/r2_sweep1.log
: > $out
for o in "" "slots_per_wave=64" "slots_per_wave=80" "sticky_pct=100" "sticky_pct=50" "slots_per_wave=64 sticky_pct=100" "slots_per_wave=64 sticky_pct=50" "schedule=0 slots_per_wave=64" ; do
  python tools/ab_bench.py c2 3 $o 2>&1 | tail -1 >> $out
done
for o in "" "slots_per_wave=64" "slots_per_wave=64 sticky_pct=25" "slots_per_wave=80" "slots_per_wave=64 sticky_pct=50"; do
  python tools/ab_bench.py c4 2 shard_rank=5 shard_world=8 $o 2>&1 | tail -1 >> $out
done
