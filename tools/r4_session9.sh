#!/bin/bash
# Round 4, GPU session 9: topped-up shading passes (tune0 = smallest top-up) on every config
root=${GRAFT_REPO_ROOT:-$PWD}; cd $root; out=gpurun_out; mkdir -p $out; tag=r4_s9
step() { local name=$1 to=$2; shift 2
    echo "== $name" | tee -a $out/$tag.log
    timeout -k 10 $to "$@" > $out/${tag}_$name.log 2>&1; local rc=$?
    echo "rc=$rc" | tee -a $out/$tag.log
    grep -h "kernel_ms_min\|frame_crc\|passed\|failed\|Error\|error" $out/${tag}_$name.log | cut -c1-260 | tail -4 | tee -a $out/$tag.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $out/$tag.log; exit 1; fi
}
step tests 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "chunked or cornell_image or tier"
for t in 0 8 16 24 32 0; do
  step c4_t$t 300 python tools/ab_bench.py c4 3 frame_out=1 tune0=$t
done
for t in 0 8 16 24 32 0; do
  step c2_t$t 200 python tools/ab_bench.py c2 4 frame_out=1 tune0=$t
done
for t in 0 16 32; do
  step c5_t$t 300 python tools/ab_bench.py c5 2 frame_out=1 tune0=$t
  step c3_t$t 200 python tools/ab_bench.py c3 3 frame_out=1 tune0=$t
done
step c4_t16_census 300 python tools/ab_bench.py c4 1 census=1 tune0=16
echo done | tee -a $out/$tag.log
