#!/bin/bash
# Round 4, GPU session 12: the slots' accumulated colour in a wave-private global area (A/B build colhbm) -> 112 / 128 slots per wave at 16 waves per CU
root=${GRAFT_REPO_ROOT:-$PWD}; cd $root; out=gpurun_out; mkdir -p $out; tag=r4_s12
step() { local name=$1 to=$2; shift 2
    echo "== $name" | tee -a $out/$tag.log
    timeout -k 10 $to "$@" > $out/${tag}_$name.log 2>&1; local rc=$?
    echo "rc=$rc" | tee -a $out/$tag.log
    grep -h "kernel_ms_min\|frame_crc\|passed\|failed\|Error\|error" $out/${tag}_$name.log | cut -c1-230 | tail -3 | tee -a $out/$tag.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $out/$tag.log; exit 1; fi
}
L=$root/owl-path-tracer_amd
PT_LIB_PATH=$L/libmi355pt_colhbm.so PT_DEFAULT_NS=128 step tests_colhbm 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cornell or chunked or tier or cube or material_coverage or c2_ or c3_mitsuba or group_walk or express"
for sc in c4 c2 c3 c5; do
  step ${sc}_product 300 python tools/ab_bench.py $sc 3 frame_out=1
  for ns in 96 112 128; do
    PT_LIB_PATH=$L/libmi355pt_colhbm.so PT_DEFAULT_NS=$ns step ${sc}_colhbm_$ns 300 python tools/ab_bench.py $sc 3 frame_out=1
  done
done
echo done | tee -a $out/$tag.log
