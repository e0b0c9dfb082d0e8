#!/bin/bash
# Round 4, GPU session 6: whole-pixel tickets first (express_mode=1) on the world-2 / world-4 / world-8 shards of C4
root=${GRAFT_REPO_ROOT:-$PWD}; cd $root; out=gpurun_out; mkdir -p $out; tag=r4_s6
step() { local name=$1 to=$2; shift 2
    echo "== $name" | tee -a $out/$tag.log
    timeout -k 10 $to "$@" > $out/${tag}_$name.log 2>&1; local rc=$?
    echo "rc=$rc" | tee -a $out/$tag.log
    grep -h "kernel_ms_min\|frame_crc\|laps_ms\|passed\|failed\|Error\|error" $out/${tag}_$name.log | cut -c1-400 | tail -6 | tee -a $out/$tag.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $out/$tag.log; exit 1; fi
}
step w2_base 300 python tools/ab_bench.py c4 2 shard_rank=1 shard_world=2 frame_out=1 timeline=1
for pm in 50 150 300 450; do
  step w2_first_$pm 300 python tools/ab_bench.py c4 2 shard_rank=1 shard_world=2 frame_out=1 express_mode=1 express_permille=$pm
done
step w2_first_300_sticky50 300 python tools/ab_bench.py c4 2 shard_rank=1 shard_world=2 frame_out=1 express_mode=1 express_permille=300 sticky_pct=50
step w2_first_300_sticky80 300 python tools/ab_bench.py c4 2 shard_rank=1 shard_world=2 frame_out=1 express_mode=1 express_permille=300 sticky_pct=80
step w4_base 300 python tools/ab_bench.py c4 2 shard_rank=1 shard_world=4 frame_out=1 timeline=1
for pm in 100 250 400 500; do
  step w4_first_$pm 300 python tools/ab_bench.py c4 2 shard_rank=1 shard_world=4 frame_out=1 express_mode=1 express_permille=$pm
done
step w8_base 300 python tools/ab_bench.py c4 2 shard_rank=1 shard_world=8 frame_out=1
step w8_ring_first_400 300 python tools/ab_bench.py c4 2 shard_rank=1 shard_world=8 frame_out=1 whole=0 express_mode=1 express_permille=400
step c4_first_100 300 python tools/ab_bench.py c4 2 frame_out=1 express_mode=1 express_permille=100
step w2_first_300_finish 300 python tools/ab_bench.py c4 1 shard_rank=1 shard_world=2 latency=1 finish=1 express_mode=1 express_permille=300
echo done | tee -a $out/$tag.log
