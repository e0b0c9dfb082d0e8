#!/bin/bash
for v in "" _ret12 _ret24 _burst2 _burst4 _keep16 _keep32 _qr2 _raylow; do
  export PT_LIB_PATH=$PWD/owl-path-tracer_amd/libmi355pt$v.so
  echo "== $v"
  python tools/ab_bench.py c4 3 | tail -1 | cut -c30-140
  python tools/ab_bench.py c3 3 | tail -1 | cut -c30-140
done
