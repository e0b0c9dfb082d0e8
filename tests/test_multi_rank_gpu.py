"""N processes, one GPU each, through the library's own communicator (pt_comm_get_unique_id / pt_comm_init_rank / pt_render): the frame
rank 0 receives must equal the single-GPU frame bit for bit.  Needs >= 2 GPUs: skipped on the one-GPU boxes of this pool (there the
same entry points run with world size 1 in tests/test_gpu_parity.py::test_library_communicator_single_rank and through
`pt_main --gpus 1`); no torch.distributed involved - the 128-byte id travels through a file, as any launcher could do it."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CHILD = r'''
import os, sys, time, numpy as np
sys.path.insert(0, %(root)r)
import ptamd; ptamd.load()
from owl_path_tracer_amd.pyhost import binding as B, scene_io
rank, world, tmp = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
dev = int(sys.argv[4]) if len(sys.argv) > 4 else rank  # (stub-collective test: every rank on device 0)
sc = scene_io.load_scene_dir(os.path.join(%(root)r, "assets"), "cornell-box")
ctx = B.Context(dev)
ctx.upload_scene(sc["entities"], [m for _, m, _ in sc["materials"]], env=B.make_env(color=(1, 1, 1), intensity=0.0))
idf = os.path.join(tmp, "comm_id.bin")
if rank == 0:
    uid = B.comm_unique_id()
    with open(idf + ".tmp", "wb") as f: f.write(uid)
    os.replace(idf + ".tmp", idf)
else:
    t0 = time.time()
    while not os.path.exists(idf):
        if time.time() - t0 > 120: raise SystemExit("no communicator id from rank 0")
        time.sleep(0.05)
    uid = open(idf, "rb").read()
ctx.comm_init_rank(uid, rank, world)
W, H = 320, 200
c = sc["camera"]
cam = B.to_camera_data(c["look_from"], c["look_at"], c["look_up"], c["vertical_fov"], W, H)
fr = B.PinnedFrame(W, H, want_rgba8=True) if rank == 0 else None
for _ in range(2):  # twice: the communicator is reused
    ctx.render_into(cam, W, H, 48, 16, fr.rgb if fr else None, fr.rgba8 if fr else None)
if rank == 0:
    np.save(os.path.join(tmp, "rgb.npy"), np.array(fr.rgb)); np.save(os.path.join(tmp, "rgba8.npy"), np.array(fr.rgba8))
ctx.comm_destroy(); ctx.close()
'''


def _device_count():
    import torch

    return torch.cuda.device_count()


def _run_ranks(tmp_path, world, env, same_device=False):
    procs = [subprocess.Popen([sys.executable, "-c", CHILD % dict(root=ROOT), str(r), str(world), str(tmp_path)] + (["0"] if same_device else []), env=env) for r in range(world)]
    t0 = time.time()
    try:
        for p in procs:
            p.wait(timeout=max(1.0, 600 - (time.time() - t0)))
    finally:  # a rank that hangs in a collective must not outlive the test holding a GPU
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    assert all(p.returncode == 0 for p in procs), [p.returncode for p in procs]
    sys.path.insert(0, ROOT)
    from owl_path_tracer_amd.pyhost import binding as B, scene_io

    sc = scene_io.load_scene_dir(os.path.join(ROOT, "assets"), "cornell-box")
    ctx = B.Context(0)
    ctx.upload_scene(sc["entities"], [m for _, m, _ in sc["materials"]], env=B.make_env(color=(1, 1, 1), intensity=0.0))
    c = sc["camera"]
    cam = B.to_camera_data(c["look_from"], c["look_at"], c["look_up"], c["vertical_fov"], 320, 200)
    want, want8 = ctx.render(cam, 320, 200, 48, 16, want_rgba8=True)
    ctx.close()
    np.testing.assert_array_equal(np.load(tmp_path / "rgb.npy").view(np.uint32), want.view(np.uint32))
    np.testing.assert_array_equal(np.load(tmp_path / "rgba8.npy"), want8)


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_library_reduce_across_processes(tmp_path, world):
    """The root asks for the RGBA8 image, the other ranks pass no buffers at all: every rank must still enqueue the same collectives
    (round-2 advisor finding: the RGBA8 reduce was issued only where a buffer was passed).  World 1 runs on the one-GPU boxes."""
    if _device_count() < world:
        pytest.skip("needs %d GPUs" % world)
    _run_ranks(tmp_path, world, dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))


@pytest.mark.parametrize("world", [2, 3, 4])  # (4 ranks + this process = 5 of the 6 processes the pool allows on the card)
def test_library_reduce_with_stub_collective(tmp_path, world):
    """The library's N-rank plumbing with N > 1 PROCESSES on ONE GPU: every rank opens device 0 and PT_RCCL_PATH points at
    tests/stub/fake_rccl.cpp (built here), whose ncclReduce is a blocking sum through files.  That is NOT RCCL and proves nothing
    about RCCL - what it exercises is our side, which had never run with more than one rank (round-3 advisor finding): the unique id
    through a file, pt_comm_init_rank per process and the pixel shard it sets, ONE reduce per frame and rank although only the root
    passes buffers, the root's RGBA8 pack of the reduced frame, a second frame on the same communicator, teardown.  The frame on
    rank 0 must be the single-GPU frame bit for bit (world 3 leaves ranks with different tile counts)."""
    import shutil

    gxx = shutil.which("g++")
    if not gxx or not os.path.exists("/opt/rocm/include/rccl/rccl.h"):
        pytest.skip("g++ or the RCCL header is missing")
    stub = str(tmp_path / "libfake_rccl.so")
    subprocess.check_call([gxx, "-O1", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", stub,
                           os.path.join(ROOT, "tests", "stub", "fake_rccl.cpp"), "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
    _run_ranks(tmp_path, world, dict(os.environ, PT_RCCL_PATH=stub), same_device=True)
