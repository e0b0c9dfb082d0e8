#!/bin/bash
# GPU suite, every config, the shards of C4, and the round's profile (tools/profile_round.sh) on one box: tools/final_session.sh <tag> <profile tag>
root=${GRAFT_REPO_ROOT:-$PWD}; cd $root; out=gpurun_out; mkdir -p $out; tag=${1:-final}; ptag=${2:-r04_final}
echo "== tests" | tee -a $out/$tag.log
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -s > $out/${tag}_tests.log 2>&1; rc=$?; echo "rc=$rc" | tee -a $out/$tag.log; tail -3 $out/${tag}_tests.log | tee -a $out/$tag.log
if [ $rc -ne 0 ]; then grep -n "Error\|FAILED\|assert" $out/${tag}_tests.log | head -20 | tee -a $out/$tag.log; fi
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
for sc in c4 c2 c3 c5; do
  echo "== $sc" | tee -a $out/$tag.log
  timeout -k 10 300 python tools/ab_bench.py $sc 3 > $out/${tag}_$sc.log 2>&1; grep -h kernel_ms_min $out/${tag}_$sc.log | cut -c1-300 | tee -a $out/$tag.log
done
for w in 2 4 8; do
  echo "== c4 world $w" | tee -a $out/$tag.log
  timeout -k 10 300 python tools/ab_bench.py c4 2 shard_rank=1 shard_world=$w > $out/${tag}_w$w.log 2>&1; grep -h kernel_ms_min $out/${tag}_w$w.log | cut -c1-300 | tee -a $out/$tag.log
done
echo "== profile" | tee -a $out/$tag.log
bash tools/profile_round.sh $ptag 2>&1 | tail -8 | tee -a $out/$tag.log
echo done | tee -a $out/$tag.log
