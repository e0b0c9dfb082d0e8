#!/bin/bash
# Collect rocprofv3 PMC passes (one counter group per run, kernel-trace only) for one ab_bench workload.
# usage: tools/pmc_run.sh <tag> <scene> <reps> -- "<group1>" "<group2>" ...     (run from the repo root on the GPU box)
set -o pipefail
tag=$1; scene=$2; reps=$3; shift 4
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
    i=$((i + 1))
    timeout -k 10 ${PMC_TIMEOUT:-90} rocprofv3 --kernel-trace --pmc $grp -d $root/gpurun_out/pmc_${tag}_$i --output-format csv -- python3 $root/tools/ab_bench.py $scene $reps $PMC_EXTRA > $root/gpurun_out/pmc_${tag}_$i.log 2>&1 || { echo "pass $i failed"; grep -m1 "error code" $root/gpurun_out/pmc_${tag}_$i.log; }
done
cd $root && python3 tools/pmc_summary.py "pt_render_wave_kernel<false" $(for j in $(seq 1 $i); do echo gpurun_out/pmc_${tag}_$j; done) > gpurun_out/pmc_${tag}_summary.json && cat gpurun_out/pmc_${tag}_summary.json
