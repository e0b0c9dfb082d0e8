#!/bin/bash
# Round 4, GPU session 18: REHEARSAL of bench.py's N = 2 path on one GPU (both ranks on device 0, stub collective): not a measurement
root=${GRAFT_REPO_ROOT:-$PWD}; cd $root; out=gpurun_out; mkdir -p $out
g++ -O1 -std=c++17 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o /tmp/libfake_rccl.so tests/stub/fake_rccl.cpp -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib || exit 1
PT_BENCH_DEVICE=0 PT_RCCL_PATH=/tmp/libfake_rccl.so timeout -k 10 800 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 > $out/r4_s18_rehearsal.log 2>&1
echo rc=$?; tail -2 $out/r4_s18_rehearsal.log | cut -c1-3000
