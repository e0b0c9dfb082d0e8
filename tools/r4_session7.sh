#!/bin/bash
# Round 4, GPU session 7: after the source split (pt_kernel.hip / pt_kernel_aux.hip / pt_schedule.hip / pt_trace.h) and the removal of the coop / quant
# paths: GPU suite; product build against the build without the lobe-bin code (alternating); upload phases on the box
root=${GRAFT_REPO_ROOT:-$PWD}; cd $root; out=gpurun_out; mkdir -p $out; tag=r4_s7
step() { local name=$1 to=$2; shift 2
    echo "== $name" | tee -a $out/$tag.log
    timeout -k 10 $to "$@" > $out/${tag}_$name.log 2>&1; local rc=$?
    echo "rc=$rc" | tee -a $out/$tag.log
    grep -h "kernel_ms_min\|frame_crc\|pt_upload_scene:\|passed\|failed\|Error\|error" $out/${tag}_$name.log | cut -c1-330 | tail -9 | tee -a $out/$tag.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $out/$tag.log; exit 1; fi
}
step tests 1100 python -m pytest tests -m gpu -x -q -s
NB=$root/owl-path-tracer_amd/libmi355pt_nobins.so
for i in 1 2; do
  step c4_bins_$i 300 python tools/ab_bench.py c4 3 frame_out=1
  PT_LIB_PATH=$NB step c4_nobins_$i 300 python tools/ab_bench.py c4 3 frame_out=1
done
step c5_bins 300 python tools/ab_bench.py c5 2 frame_out=1
PT_LIB_PATH=$NB step c5_nobins 300 python tools/ab_bench.py c5 2 frame_out=1
step c2_bins 200 python tools/ab_bench.py c2 4 frame_out=1
PT_LIB_PATH=$NB step c2_nobins 200 python tools/ab_bench.py c2 4 frame_out=1
PT_UPLOAD_TRACE=1 step c4_upload 300 python tools/ab_bench.py c4 1
PT_UPLOAD_TRACE=1 step c5_upload 300 python tools/ab_bench.py c5 1
echo done | tee -a $out/$tag.log
