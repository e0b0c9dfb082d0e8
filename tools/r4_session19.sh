#!/bin/bash
# Round 4, GPU session 19: slab distances by packed fma (A/B build boxfma) against the product build
root=${GRAFT_REPO_ROOT:-$PWD}; cd $root; out=gpurun_out; mkdir -p $out; tag=r4_s19
step() { local name=$1 to=$2; shift 2
    echo "== $name" | tee -a $out/$tag.log
    timeout -k 10 $to "$@" > $out/${tag}_$name.log 2>&1; local rc=$?
    echo "rc=$rc" | tee -a $out/$tag.log
    grep -h "kernel_ms_min\|frame_crc\|\"rays\"\|passed\|failed\|Error\|error" $out/${tag}_$name.log | cut -c1-230 | tail -3 | tee -a $out/$tag.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $out/$tag.log; exit 1; fi
}
L=$root/owl-path-tracer_amd
PT_LIB_PATH=$L/libmi355pt_boxfma.so step tests_boxfma 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not fallback_build and not lobe_bins_build"
for rep in 1 2; do
  for sc in c4 c2 c3 c5; do
    step ${sc}_product_$rep 300 python tools/ab_bench.py $sc 3 frame_out=1 shard_rank=0 shard_world=1
    PT_LIB_PATH=$L/libmi355pt_boxfma.so step ${sc}_boxfma_$rep 300 python tools/ab_bench.py $sc 3 frame_out=1 shard_rank=0 shard_world=1
  done
done
echo done | tee -a $out/$tag.log
