#!/bin/bash
# Round 4, GPU session 3: traversal census (stack entries beyond the best hit), threaded builder timing, sanity of C4 after the lobe-bin code went in (default off)
root=${GRAFT_REPO_ROOT:-$PWD}; cd $root; out=gpurun_out; mkdir -p $out; tag=r4_s3
step() { local name=$1 to=$2; shift 2
    echo "== $name" | tee -a $out/$tag.log
    timeout -k 10 $to "$@" > $out/${tag}_$name.log 2>&1; local rc=$?
    echo "rc=$rc" | tee -a $out/$tag.log
    grep -h "kernel_ms_min\|passed\|failed\|Error\|error" $out/${tag}_$name.log | cut -c1-420 | tail -6 | tee -a $out/$tag.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $out/$tag.log; exit 1; fi
}
step c4 300 python tools/ab_bench.py c4 3 census=1 frame_out=1
step c5 300 python tools/ab_bench.py c5 2 census=1 frame_out=1
step c2 200 python tools/ab_bench.py c2 4 census=1 frame_out=1
step c3 200 python tools/ab_bench.py c3 3 census=1 frame_out=1
PT_BUILD_THREADS=1 step c4_build1 300 python tools/ab_bench.py c4 1 frame_out=1
step tests 1100 python -m pytest tests -m gpu -x -q -s
echo done | tee -a $out/$tag.log
