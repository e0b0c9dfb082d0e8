#!/bin/bash
# round-3 evidence run on one box: bench profile (stats + PMC), numbers of the other configs / shards, PMC of the 1/8 shard and of C2
tag=${1:-r03a}
bash tools/profile_round.sh $tag > gpurun_out/prof_$tag.log 2>&1
tail -3 gpurun_out/prof_$tag.log
bash tools/round_numbers.sh $tag > /dev/null 2>&1
cat gpurun_out/numbers_$tag.log | cut -c1-300
PMC_EXTRA="shard_rank=3 shard_world=8" bash tools/pmc_run.sh ${tag}_shard8 c4 1 -- "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum" > gpurun_out/pmc_${tag}_shard8.log 2>&1
PMC_EXTRA="" bash tools/pmc_run.sh ${tag}_c2 c2 1 -- "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum" > gpurun_out/pmc_${tag}_c2.log 2>&1
grep -h "valu_lane_utilization\|l2_hit_rate" gpurun_out/pmc_${tag}_shard8_summary.json gpurun_out/pmc_${tag}_c2_summary.json
