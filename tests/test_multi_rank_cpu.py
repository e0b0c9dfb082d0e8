"""World-size-2 CPU test of the multi-GPU path (gloo): product sharding (pt_shard_pixels, C-ABI) + the single reduce of
pyhost.distributed.  No GPU here, so each rank renders its shard with the ORACLE (checker standing in for the HIP render);
what is under test is the host logic that bench.py runs over RCCL: disjoint tiles, zero elsewhere, exact sum."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch
ROOT = sys.argv[1]
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import ptamd; ptamd.load()
from owl_path_tracer_amd.pyhost import binding as B, distributed as D, scene_io
import oracle as orc

rank, local_rank, world = D.init(backend="gloo")
W, H, spp, depth = 72, 40, 6, 8
sc = scene_io.load_scene_dir(os.path.join(ROOT, "assets"), "cube")
flat = scene_io.flatten_scene(sc["entities"], sc["materials"], {0: scene_io.checker_texture()})
S = orc.Scene(flat)
c = sc["camera"]
cam = orc.camera_from_array(B.to_camera_data(c["look_from"], c["look_at"], c["look_up"], c["vertical_fov"], W, H).as_array())
env = orc.make_env(use_auto=True, intensity=1.0)
ids = B.shard_pixels(W, H, D.TILE, rank, world)          # product code: which pixels this rank owns
part = np.zeros((H, W, 3), np.float32)
S.render(cam, env, W, H, spp, depth, pixel_list=ids, out=part, threads=2)
own = D.owned_mask(B, W, H, rank, world).numpy()
assert not part[~own].any(), "a rank wrote outside its tiles"
fb = torch.from_numpy(part)
D.reduce_framebuffer(fb, dst=0)                            # the one collective
if rank == 0:
    full, _, _ = S.render(cam, env, W, H, spp, depth, threads=2)
    same = fb.numpy().view(np.uint32) == full.view(np.uint32)
    assert same.all(), "sum over ranks differs from the single-rank image in %d floats" % (~same).sum()
    print("MULTI_RANK_OK", world, int(ids.size))
torch.distributed.barrier()
torch.distributed.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_render_reduces_to_single_rank_image(tmp_path, world):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = 29500 + (os.getpid() % 2000) + world
    env = dict(os.environ, OMP_NUM_THREADS="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script), ROOT]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "MULTI_RANK_OK %d" % world in out.stdout
