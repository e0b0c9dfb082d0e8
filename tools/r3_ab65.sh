#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
show() {
python3 - "$1" <<'PY'
import csv,glob,sys
fs=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv', recursive=True)
rows=list(csv.DictReader(open(fs[0])))
for r in rows[-10:]:
    print(r['Kernel_Name'][:44], round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6,3), r.get('Grid_Size_X', r.get('Grid_Size','')))
PY
}
for o in "whole=0" "whole=-1"; do
  rm -rf $root/gpurun_out/tr_c2
  timeout -k 10 120 rocprofv3 --kernel-trace -d $root/gpurun_out/tr_c2 --output-format csv -- python3 $root/tools/ab_bench.py c2 3 $o > /dev/null 2>&1
  echo "== c2 $o"; show $root/gpurun_out/tr_c2
done
rm -rf $root/gpurun_out/tr_c2
timeout -k 10 120 rocprofv3 --kernel-trace -d $root/gpurun_out/tr_c2 --output-format csv -- python3 $root/tools/ab_bench.py c4 2 shard_rank=1 shard_world=8 > /dev/null 2>&1
echo "== shard8"; show $root/gpurun_out/tr_c2
rm -rf $root/gpurun_out/tr_c2
