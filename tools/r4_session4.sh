#!/bin/bash
# Round 4, GPU session 4: quad nodes collapsed by surface area against the fixed two-level collapse (host-side change only)
root=${GRAFT_REPO_ROOT:-$PWD}; cd $root; out=gpurun_out; mkdir -p $out; tag=r4_s4
step() { local name=$1 to=$2; shift 2
    echo "== $name" | tee -a $out/$tag.log
    timeout -k 10 $to "$@" > $out/${tag}_$name.log 2>&1; local rc=$?
    echo "rc=$rc" | tee -a $out/$tag.log
    grep -h "kernel_ms_min\|frame_crc\|\"rays\"\|passed\|failed\|Error\|error" $out/${tag}_$name.log | cut -c1-330 | tail -6 | tee -a $out/$tag.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $out/$tag.log; exit 1; fi
}
for sc in c4 c5 c2 c3; do
  for m in 0 1; do
    PT_COLLAPSE4_BY_AREA=$m step ${sc}_area$m 300 python tools/ab_bench.py $sc 3 frame_out=1 shard_rank=0 shard_world=1
  done
done
PT_COLLAPSE4_BY_AREA=0 step c4_area0_again 300 python tools/ab_bench.py c4 3 frame_out=1
PT_COLLAPSE4_BY_AREA=1 step c4_area1_again 300 python tools/ab_bench.py c4 3 frame_out=1
step tests 1100 python -m pytest tests -m gpu -x -q -s
echo done | tee -a $out/$tag.log
