#!/bin/bash
# Round 4, GPU session 10: top-up threshold (default 8 now) against off / 1 / 4; strictly distance-sorted pushes x quad collapse mode
root=${GRAFT_REPO_ROOT:-$PWD}; cd $root; out=gpurun_out; mkdir -p $out; tag=r4_s10
step() { local name=$1 to=$2; shift 2
    echo "== $name" | tee -a $out/$tag.log
    timeout -k 10 $to "$@" > $out/${tag}_$name.log 2>&1; local rc=$?
    echo "rc=$rc" | tee -a $out/$tag.log
    grep -h "kernel_ms_min\|frame_crc\|\"rays\"\|passed\|failed\|Error\|error" $out/${tag}_$name.log | cut -c1-260 | tail -4 | tee -a $out/$tag.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $out/$tag.log; exit 1; fi
}
for t in 0 255 1 4; do
  step c4_t$t 300 python tools/ab_bench.py c4 3 frame_out=1 tune0=$t
  step c2_t$t 200 python tools/ab_bench.py c2 4 frame_out=1 tune0=$t
  step c3_t$t 200 python tools/ab_bench.py c3 3 frame_out=1 tune0=$t
done
SO=$root/owl-path-tracer_amd/libmi355pt_sorted.so
for sc in c4 c5 c2; do
  for m in 0 1; do
    PT_COLLAPSE4_BY_AREA=$m step ${sc}_area${m}_heur 300 python tools/ab_bench.py $sc 2 frame_out=1 shard_rank=0 shard_world=1
    PT_COLLAPSE4_BY_AREA=$m PT_LIB_PATH=$SO step ${sc}_area${m}_sorted 300 python tools/ab_bench.py $sc 2 frame_out=1 shard_rank=0 shard_world=1
  done
done
PT_COLLAPSE4_BY_AREA=1 step c5_area1_census 300 python tools/ab_bench.py c5 1 census=1
echo done | tee -a $out/$tag.log
