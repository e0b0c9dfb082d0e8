#!/bin/bash
# generic same-box A/B: AB_TAG, AB_VARIANTS (space separated, "-" = product build), AB_SCENES, AB_REPS
root=${GRAFT_REPO_ROOT:-$PWD}; cd $root; out=gpurun_out; mkdir -p $out; tag=${AB_TAG:-ab}
L=$root/owl-path-tracer_amd
for rep in $(seq 1 ${AB_REPS:-2}); do
  for v in $AB_VARIANTS; do
    lib=$L/libmi355pt_$v.so; [ "$v" = "-" ] && lib=$L/libmi355pt.so
    for sc in ${AB_SCENES:-c4 c2}; do
      r=$(PT_LIB_PATH=$lib timeout -k 10 300 python tools/ab_bench.py $sc 3 2>&1 | grep kernel_ms_min | sed 's/.*"kernel_ms_min": \([0-9.]*\), "kernel_ms_med": \([0-9.]*\).*/\1 \2/')
      echo "$sc $v rep$rep: $r" | tee -a $out/$tag.log
    done
  done
done
echo done | tee -a $out/$tag.log
