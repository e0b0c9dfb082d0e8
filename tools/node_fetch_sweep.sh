#!/bin/bash
# sweep of tools/node_fetch_bench on the GPU box: modes x table sizes x active lanes x occupancy
cd "$(dirname "$0")"
for kib in 2048 16384 65536; do
  for mode in 0 1 2 3; do
    for act in 64 32 8; do
      for pad in 9 17; do
        ./node_fetch_bench $mode $kib 512 $act $pad
      done
    done
  done
done
