#!/bin/bash
for lib in libmi355pt.so libmi355pt_rot.so; do
  export PT_LIB_PATH=$PWD/owl-path-tracer_amd/$lib
  echo "== $lib"
  python tools/ab_bench.py c4 3 census=1 | grep -v node_steps | cut -c1-260
  python tools/ab_bench.py c3 3 | tail -1 | cut -c1-200
done
