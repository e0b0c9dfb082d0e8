#!/bin/bash
# Round 4, GPU session 5: C2 slots-per-wave sweep beyond 104; finish-time diagnostics of the world-2 / world-4 / world-8 shards of C4
root=${GRAFT_REPO_ROOT:-$PWD}; cd $root; out=gpurun_out; mkdir -p $out; tag=r4_s5
step() { local name=$1 to=$2; shift 2
    echo "== $name" | tee -a $out/$tag.log
    timeout -k 10 $to "$@" > $out/${tag}_$name.log 2>&1; local rc=$?
    echo "rc=$rc" | tee -a $out/$tag.log
    grep -h "kernel_ms_min\|passed\|failed\|Error\|error" $out/${tag}_$name.log | cut -c1-330 | tail -6 | tee -a $out/$tag.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $out/$tag.log; exit 1; fi
}
for ns in 0 96 128 160 192 224 252; do
  step c2_ns$ns 200 python tools/ab_bench.py c2 4 frame_out=1 slots_per_wave=$ns
done
for ns in 128 160; do
  step c3_ns$ns 200 python tools/ab_bench.py c3 3 frame_out=1 slots_per_wave=$ns
done
for w in 2 4 8; do
  step w${w} 300 python tools/ab_bench.py c4 2 shard_rank=1 shard_world=$w frame_out=1
  step w${w}_finish 300 python tools/ab_bench.py c4 1 shard_rank=1 shard_world=$w latency=1 finish=1 tiers=1
done
echo done | tee -a $out/$tag.log
