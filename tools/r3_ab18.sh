#!/bin/bash
# what does a ray cost in a wave that has few of them, on an idle machine?  (1/512 shard of C4: 1536 pixels)
for o in "groups=2 slots_per_wave=16" "groups=2 slots_per_wave=64" "groups=0 slots_per_wave=64" "groups=1"; do
  echo "== $o"
  python tools/ab_bench.py c4 2 shard_rank=5 shard_world=512 census=1 chain=1 express_permille=0 $o | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if 'node_steps' in d:
        cyc = d['cycle_share']
        print('cycle_share', cyc); print('groups', d['groups'])
        print('hit passes', d['hit_passes'], 'items/pass', d['hit_items/hit_passes'], 'miss passes', d['miss_passes'], 'items', d['miss_items/miss_passes'], 'rays', d['rays'], 'node_steps', d['node_steps'], 'tri_steps', d['tri_steps'], 'retire_passes', d['retire_passes'])
    elif 'chain' in d: print(d)
    elif 'kernel_ms_min' in d: print(d['kernel_ms_min'], d['grid'])
"
done
