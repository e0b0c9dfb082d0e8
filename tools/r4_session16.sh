#!/bin/bash
# Round 4, GPU session 16: the tree as it will be judged - smoke(), GPU suite, bench.py with the driver's arguments
root=${GRAFT_REPO_ROOT:-$PWD}; cd $root; out=gpurun_out; mkdir -p $out; tag=r4_s16
echo "== smoke" | tee -a $out/$tag.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.log 2>&1; rc=$?; echo "rc=$rc" | tee -a $out/$tag.log; tail -2 $out/${tag}_smoke.log | tee -a $out/$tag.log
[ $rc -eq 124 ] && exit 1
echo "== tests" | tee -a $out/$tag.log
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -s > $out/${tag}_tests.log 2>&1; rc=$?; echo "rc=$rc" | tee -a $out/$tag.log; tail -3 $out/${tag}_tests.log | tee -a $out/$tag.log; grep "C3 sweep\|hand-off\|kernel_ms=" $out/${tag}_tests.log | tee -a $out/$tag.log
[ $rc -eq 124 ] && exit 1
echo "== bench" | tee -a $out/$tag.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench.log 2>&1; rc=$?; echo "rc=$rc" | tee -a $out/$tag.log; tail -1 $out/${tag}_bench.log | cut -c1-3000 | tee -a $out/$tag.log
echo done | tee -a $out/$tag.log
