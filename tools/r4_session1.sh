#!/bin/bash
# Round 4, GPU session 1: the GPU suite, then product-instance time + scheduler / lobe census of C5, C2, C3, C4, the C3 material sweep and PMC of C5.
# A step that is killed by its timeout ends the session (no further GPU step after a hang).
root=${GRAFT_REPO_ROOT:-$PWD}; cd $root; out=gpurun_out; mkdir -p $out
step() { # name timeout cmd...
    local name=$1 to=$2; shift 2
    echo "== $name" | tee -a $out/r4_s1.log
    timeout -k 10 $to "$@" > $out/r4_s1_$name.log 2>&1; local rc=$?
    echo "rc=$rc" | tee -a $out/r4_s1.log
    tail -3 $out/r4_s1_$name.log | cut -c1-400
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $out/r4_s1.log; exit 1; fi
}
step tests 1100 python -m pytest tests -m gpu -x -q -s
step c5 300 python tools/ab_bench.py c5 2 census=1
step c2 200 python tools/ab_bench.py c2 3 census=1
step c3 200 python tools/ab_bench.py c3 3 census=1
step c4 300 python tools/ab_bench.py c4 2 census=1
step c3sweep 400 python tools/ab_bench.py c3 2 sweep=1
PMC_TIMEOUT=240 step pmc_c5 1500 bash tools/pmc_run.sh r4c5 c5 1 -- "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum"
echo done | tee -a $out/r4_s1.log
