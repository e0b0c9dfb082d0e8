#!/bin/bash
out=gpurun_out/r2_sweep11.log
: > $out
python -m pytest tests -x -q -m gpu -k "closest_hit or cube_image or cornell_image or material_coverage or chunked or one_quad" > gpurun_out/r2_sweep11_tests.log 2>&1 || { tail -30 gpurun_out/r2_sweep11_tests.log; exit 1; }
tail -1 gpurun_out/r2_sweep11_tests.log
V=owl-path-tracer_amd/variants
for lib in owl-path-tracer_amd/libmi355pt.so $V/lib_nobf.so owl-path-tracer_amd/libmi355pt.so $V/lib_nobf.so; do
  PT_LIB_PATH=$lib python tools/ab_bench.py c4 2 2>&1 | tail -1 >> $out
  PT_LIB_PATH=$lib python tools/ab_bench.py c4 2 shard_rank=5 shard_world=8 2>&1 | tail -1 >> $out
  PT_LIB_PATH=$lib python tools/ab_bench.py c2 3 2>&1 | tail -1 >> $out
done
