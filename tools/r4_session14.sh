#!/bin/bash
# Round 4, GPU session 14: the round's numbers on one box (every config, shards of C4 with chain lengths) + C5 leaf-size sweep
root=${GRAFT_REPO_ROOT:-$PWD}; cd $root; out=gpurun_out; mkdir -p $out; tag=r4_s14
step() { local name=$1 to=$2; shift 2
    echo "== $name" | tee -a $out/$tag.log
    timeout -k 10 $to "$@" > $out/${tag}_$name.log 2>&1; local rc=$?
    echo "rc=$rc" | tee -a $out/$tag.log
    grep -h "kernel_ms_min\|\"chain\"\|passed\|failed\|Error\|error" $out/${tag}_$name.log | cut -c1-420 | tail -3 | tee -a $out/$tag.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $out/$tag.log; exit 1; fi
}
step c4 300 python tools/ab_bench.py c4 3 chain=1
for w in 2 4 8 16 64; do
  step c4_w$w 300 python tools/ab_bench.py c4 2 shard_rank=1 shard_world=$w chain=1
done
step c4_w512 300 python tools/ab_bench.py c4 2 shard_rank=1 shard_world=512
step c2 200 python tools/ab_bench.py c2 4
step c3 200 python tools/ab_bench.py c3 3
step c5 300 python tools/ab_bench.py c5 2
for ls in 3 6; do
  step c5_leaf$ls 300 python tools/ab_bench.py c5 2 leaf_size=$ls
done
step c5_w8 300 python tools/ab_bench.py c5 2 shard_rank=1 shard_world=8
echo done | tee -a $out/$tag.log
