#!/bin/bash
out=gpurun_out/r2_gather.log
: > $out
for kib in 16 256 2048 8192 32768 65536 131072; do
  for lanes in 64 32; do
    ./tools/fetch_calib -$kib 256 $lanes >> $out
  done
done
