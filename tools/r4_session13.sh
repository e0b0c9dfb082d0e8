#!/bin/bash
# Round 4, GPU session 13: the round's profile of the bench command (kernel-trace stats + PMC passes), GPU suite, PMC of C5 / C2 / world-8 shard on the final build
root=${GRAFT_REPO_ROOT:-$PWD}; cd $root; out=gpurun_out; mkdir -p $out; tag=r4_s13
echo "== tests" | tee -a $out/$tag.log
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -s > $out/${tag}_tests.log 2>&1; rc=$?; echo "rc=$rc" | tee -a $out/$tag.log; tail -3 $out/${tag}_tests.log | tee -a $out/$tag.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
echo "== profile" | tee -a $out/$tag.log
bash tools/profile_round.sh r04_final 2>&1 | tail -12 | tee -a $out/$tag.log
echo done | tee -a $out/$tag.log
