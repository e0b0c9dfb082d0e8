#!/bin/bash
# Round 4, GPU session 8: product build without the lobe-bin code; GPU suite (incl. the lobebins build's parity test); timings of every config; upload phases
root=${GRAFT_REPO_ROOT:-$PWD}; cd $root; out=gpurun_out; mkdir -p $out; tag=r4_s8
step() { local name=$1 to=$2; shift 2
    echo "== $name" | tee -a $out/$tag.log
    timeout -k 10 $to "$@" > $out/${tag}_$name.log 2>&1; local rc=$?
    echo "rc=$rc" | tee -a $out/$tag.log
    grep -h "kernel_ms_min\|frame_crc\|pt_upload_scene:\|passed\|failed\|Error\|error" $out/${tag}_$name.log | cut -c1-400 | tail -9 | tee -a $out/$tag.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $out/$tag.log; exit 1; fi
}
step tests 1100 python -m pytest tests -m gpu -x -q -s
PT_UPLOAD_TRACE=1 step c4 300 python tools/ab_bench.py c4 3 frame_out=1
PT_UPLOAD_TRACE=1 step c5 300 python tools/ab_bench.py c5 2 frame_out=1
step c2 200 python tools/ab_bench.py c2 4 frame_out=1
step c3 200 python tools/ab_bench.py c3 3 frame_out=1
echo done | tee -a $out/$tag.log
