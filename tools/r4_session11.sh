#!/bin/bash
# Round 4, GPU session 11: a shading pass as soon as both queues hold a full batch together (A/B builds comb8 / comb16) against the product build
root=${GRAFT_REPO_ROOT:-$PWD}; cd $root; out=gpurun_out; mkdir -p $out; tag=r4_s11
step() { local name=$1 to=$2; shift 2
    echo "== $name" | tee -a $out/$tag.log
    timeout -k 10 $to "$@" > $out/${tag}_$name.log 2>&1; local rc=$?
    echo "rc=$rc" | tee -a $out/$tag.log
    grep -h "kernel_ms_min\|frame_crc\|passed\|failed\|Error\|error" $out/${tag}_$name.log | cut -c1-200 | tail -3 | tee -a $out/$tag.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $out/$tag.log; exit 1; fi
}
L=$root/owl-path-tracer_amd
for rep in 1 2; do
  for v in "" _comb8 _comb16; do
    PT_LIB_PATH=$L/libmi355pt$v.so step c4${v}_$rep 300 python tools/ab_bench.py c4 3 frame_out=1
    PT_LIB_PATH=$L/libmi355pt$v.so step c2${v}_$rep 200 python tools/ab_bench.py c2 4 frame_out=1
    PT_LIB_PATH=$L/libmi355pt$v.so step c3${v}_$rep 200 python tools/ab_bench.py c3 3 frame_out=1
  done
done
echo done | tee -a $out/$tag.log
