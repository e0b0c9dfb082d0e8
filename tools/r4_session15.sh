#!/bin/bash
# Round 4, GPU session 15: more express pixels than an eighth of the waves (express_cap) with the grid oversubscribed by the express waves (express_over)
root=${GRAFT_REPO_ROOT:-$PWD}; cd $root; out=gpurun_out; mkdir -p $out; tag=r4_s15
step() { local name=$1 to=$2; shift 2
    echo "== $name" | tee -a $out/$tag.log
    timeout -k 10 $to "$@" > $out/${tag}_$name.log 2>&1; local rc=$?
    echo "rc=$rc" | tee -a $out/$tag.log
    grep -h "kernel_ms_min\|frame_crc\|passed\|failed\|Error\|error" $out/${tag}_$name.log | cut -c1-330 | tail -2 | tee -a $out/$tag.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $out/$tag.log; exit 1; fi
}
for w in 2 4; do
  step w${w}_base 300 python tools/ab_bench.py c4 2 shard_rank=1 shard_world=$w frame_out=1
  step w${w}_over_only 300 python tools/ab_bench.py c4 2 shard_rank=1 shard_world=$w frame_out=1 express_over=1
  step w${w}_A 300 python tools/ab_bench.py c4 2 shard_rank=1 shard_world=$w frame_out=1 express_permille=20 ns_express=16 express_cap=250 express_over=1
  step w${w}_B 300 python tools/ab_bench.py c4 2 shard_rank=1 shard_world=$w frame_out=1 express_permille=50 ns_express=32 express_cap=400 express_over=1
  step w${w}_C 300 python tools/ab_bench.py c4 2 shard_rank=1 shard_world=$w frame_out=1 express_permille=100 ns_express=32 express_cap=500 express_over=1
  step w${w}_D 300 python tools/ab_bench.py c4 2 shard_rank=1 shard_world=$w frame_out=1 express_permille=50 ns_express=32 express_cap=400 express_over=0
  step w${w}_E 300 python tools/ab_bench.py c4 2 shard_rank=1 shard_world=$w frame_out=1 express_permille=10 ns_express=8 express_cap=250 express_over=1
  step w${w}_F 300 python tools/ab_bench.py c4 2 shard_rank=1 shard_world=$w frame_out=1 express_permille=150 ns_express=48 express_cap=600 express_over=1
done
echo done | tee -a $out/$tag.log
