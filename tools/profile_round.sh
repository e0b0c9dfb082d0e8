#!/bin/bash
# Round profile of the bench command on one GPU box: kernel-trace stats + PMC passes (each its own run).  usage: tools/profile_round.sh <tag>
set -o pipefail
tag=$1
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/stats --output-format csv -- python3 $root/bench.py --steps 3 --warmup 1 > $out/bench.log 2>&1 || { echo "stats pass failed"; tail -5 $out/bench.log; exit 1; }
tail -1 $out/bench.log
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TA_TA_BUSY_sum TD_TD_BUSY_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM"; do
    i=$((i + 1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d $out/pmc_$i --output-format csv -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_$i.log 2>&1 || { echo "pmc pass $i failed"; grep -m1 "error code" $out/pmc_$i.log; }
done
cd $root && python3 tools/pmc_summary.py "pt_render_wave_kernel<false" $(for j in $(seq 1 $i); do echo $out/pmc_$j; done) > $out/pmc_summary.json
find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
find $out/stats -name "*kernel_trace.csv" -exec cp {} $out/kernel_trace.csv \;
rm -rf $out/stats
head -5 $out/kernel_stats.csv
