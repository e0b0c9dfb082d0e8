#!/bin/bash
# Round 4, GPU session 22: same-box A/B of the slab-test forms: product (fma + run-time switch to the subtracting form), fma only, subtracting only
root=${GRAFT_REPO_ROOT:-$PWD}; cd $root; out=gpurun_out; mkdir -p $out; tag=r4_s22
L=$root/owl-path-tracer_amd
for rep in 1 2 3; do
  for v in "" _fmaonly _subonly; do
    for sc in c4 c2; do
      echo "== ${sc}${v} rep $rep" | tee -a $out/$tag.log
      PT_LIB_PATH=$L/libmi355pt$v.so timeout -k 10 300 python tools/ab_bench.py $sc 3 2>&1 | grep kernel_ms_min | cut -c1-170 | tee -a $out/$tag.log
    done
  done
done
echo done | tee -a $out/$tag.log
