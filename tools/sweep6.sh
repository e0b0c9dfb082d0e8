#!/bin/bash
out=gpurun_out/r2_gather2.log
: > $out
for kib in 2048 65536; do for r in 64 128 192 256; do ./tools/gather_rec_$r $kib 256 >> $out; done; done
