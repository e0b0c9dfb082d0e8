#!/bin/bash
# One-box set of the round's numbers: chain turnaround at several loads (tools/ab_bench.py chain=1), C2, and the full-size configs
# C3/C5 as the GPU parity tests print them.  usage: tools/round_numbers.sh <tag>
tag=$1
out=gpurun_out/numbers_$tag.log
: > $out
for o in "" "shard_rank=1 shard_world=2" "shard_rank=1 shard_world=4" "shard_rank=5 shard_world=8" "shard_rank=5 shard_world=64" "shard_rank=5 shard_world=512"; do
  python tools/ab_bench.py c4 3 chain=1 $o 2>&1 | tail -2 >> $out
done
python tools/ab_bench.py c2 3 chain=1 2>&1 | tail -2 >> $out
python -m pytest tests/test_gpu_parity.py -q -s -m gpu -k "c3_mitsuba or c5_car or c2_cornell or c4_dragon" 2>&1 | grep "kernel_ms\|passed\|failed" >> $out
cat $out | cut -c1-400
