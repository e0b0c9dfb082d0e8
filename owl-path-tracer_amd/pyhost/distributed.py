"""Multi-GPU glue for bench.py and the multi-rank tests: one process per GPU, pixel tiles sharded over ranks, ONE reduce.

The reference is single-GPU (create_context(nullptr, 1), path_tracer/src/application.cpp:62).  Pixels are independent
(per-pixel RNG seeded from (x, y), device.cu:226) but the samples of one pixel are not (the stream is carried across samples,
device.cu:229-243), so the frame is sharded by pixel tile, never by sample.  Every rank renders the tiles
`pt_shard_pixels(W, H, tile, rank, world)` into a zero-initialised W*H*3 float buffer; because every pixel has exactly one
non-zero contributor the sum over ranks is exact, i.e. the N-GPU image is bit-identical to the 1-GPU image.
On the GPU the reduce is the LIBRARY's (pt_comm_init_rank + pt_render / pt_group_render: RCCL inside libmi355pt.so, pt_comm.cpp);
bench.py uses torch.distributed - over gloo, on the CPU - only for the rendezvous (the 128-byte communicator id), the barrier, the
max over ranks and the per-rank timing table: the library's RCCL communicator is the only one in the process.
`reduce_framebuffer` below is the same sum over gloo for the CPU tests of the sharding logic (tests/test_multi_rank_cpu.py), where
no GPU and therefore no RCCL exists.
"""
import os

import torch
import torch.distributed as dist

TILE = 16


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None, device=None):
    """Join the process group described by RANK/WORLD_SIZE/MASTER_* (no-op for a single rank)."""
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend or "gloo")
    return rank, local_rank, world


def reduce_framebuffer(fb, dst=0):
    """CPU-test stand-in (gloo) for the library's RCCL reduce: sum of the float3 framebuffer to rank `dst`."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(fb, dst=dst, op=dist.ReduceOp.SUM)
    return fb


def owned_mask(binding, W, H, rank, world, tile=TILE):
    """Boolean (H, W) mask in FRAMEBUFFER order (row 0 = top, device.cu:251) of the pixels `rank` renders."""
    ids = binding.shard_pixels(W, H, tile, rank, world)
    m = torch.zeros(W * H, dtype=torch.bool)
    m[torch.from_numpy(ids.astype("int64"))] = True
    return m.view(H, W).flip(0)
