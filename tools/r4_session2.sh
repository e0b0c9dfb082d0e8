#!/bin/bash
# Round 4, GPU session 2: lobe-coherent hit passes - GPU suite, then A/B (bins off / on, pure_min sweep) on C2, C5, C3 sweep, C4/C3 sanity.
root=${GRAFT_REPO_ROOT:-$PWD}; cd $root; out=gpurun_out; mkdir -p $out; tag=r4_s2
step() { # name timeout cmd...
    local name=$1 to=$2; shift 2
    echo "== $name" | tee -a $out/$tag.log
    timeout -k 10 $to "$@" > $out/${tag}_$name.log 2>&1; local rc=$?
    echo "rc=$rc" | tee -a $out/$tag.log
    grep -h "kernel_ms_min\|passed\|failed\|Error\|error" $out/${tag}_$name.log | cut -c1-300 | tail -12 | tee -a $out/$tag.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $out/$tag.log; exit 1; fi
}
step tests 1100 python -m pytest tests -m gpu -x -q -s
for v in "lobe_bins=0" "lobe_bins=1" "lobe_bins=1 tune4=16" "lobe_bins=1 tune4=32" "lobe_bins=1 tune4=48" "lobe_bins=0"; do
    step "c2_$(echo $v | tr ' =' '__')" 200 python tools/ab_bench.py c2 4 frame_out=1 $v
done
for v in "lobe_bins=0" "lobe_bins=1" "lobe_bins=1 tune4=16" "lobe_bins=1 tune4=40"; do
    step "c5_$(echo $v | tr ' =' '__')" 300 python tools/ab_bench.py c5 2 frame_out=1 $v
done
step c5_census 300 python tools/ab_bench.py c5 1 census=1
step c2_census 200 python tools/ab_bench.py c2 2 census=1
step c3sweep_off 400 python tools/ab_bench.py c3 2 sweep=1 lobe_bins=0
step c3sweep_on 400 python tools/ab_bench.py c3 2 sweep=1 lobe_bins=1
step c4 300 python tools/ab_bench.py c4 3 frame_out=1
step c3 200 python tools/ab_bench.py c3 3 frame_out=1
echo done | tee -a $out/$tag.log
