"""CPU-side checks of the product library (no GPU needed): the C-ABI loads and exports every symbol declared in
include/mi355pt.h, the host BVH builder agrees with brute force, the camera and pixel-shard helpers are right,
and the render path refuses to run without the GPU (no fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from owl_path_tracer_amd.pyhost import binding as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(B.HEADER_PATH).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = C.CDLL(B.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 17
    for n in names:
        assert hasattr(L, n), "missing export " + n
    assert sorted(B.EXPORTS) == names
    assert B.lib().pt_abi_version() == 5


def test_create_without_gpu_fails_loudly_or_succeeds_on_gfx950():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present; covered by the gpu suite")
    with pytest.raises(B.PtError, match="no usable HIP device|gfx950"):
        B.Context(0)


def test_host_only_context_has_no_render_fallback(cube):
    ctx = B.Context(-1)
    ctx.upload_scene(cube["entities"], [m for _, m, _ in cube["materials"]], textures=[np.zeros((2, 2), np.uint32)], mesh_textures=[0])
    cam = B.to_camera_data([2, 1, 2], [0, 0, 0], [0, 1, 0], 50, 8, 8)
    with pytest.raises(B.PtError, match="no CPU fallback"):
        ctx.render(cam, 8, 8, 1, 4)
    with pytest.raises(B.PtError, match="needs the GPU"):
        ctx.debug_eval("sin", np.zeros(4, np.float32), 1)


def test_upload_validation(cube):
    ctx = B.Context(-1)
    ents = cube["entities"]
    mats = [m for _, m, _ in cube["materials"]]
    bad = dict(ents[0][0], normals=np.zeros((0, 3), np.float32))
    with pytest.raises(B.PtError, match="no normal"):  # the reference traps here (macros.hpp:5-11)
        ctx.upload_scene([(bad, 0)], mats)
    with pytest.raises(B.PtError, match="material index"):
        ctx.upload_scene([(ents[0][0], 5)], mats)
    with pytest.raises(B.PtError, match="before pt_upload_scene|NO_SCENE|pt_set_materials"):
        B.Context(-1).set_materials(mats)


def test_camera_matches_oracle_bitwise(orc):
    for args in (([3, 1, 0], [0, 1, 0], [0, 1, 0], 50, 512, 512), ([4, 2.5, 0], [0, .75, 0], [0, 1, 0], 50, 1920, 1080),
                 ([2, 1, 2], [0, 0, 0], [0, 1, 0], 50, 256, 256), ([0, 2, 5], [0, .5, 0], [0, 1, 0], 45, 1080, 1440)):
        a = B.to_camera_data(*args).as_array()
        b = orc.to_camera_data(*args).as_array()
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("leaf", [1, 4, 7])
def test_product_bvh_equals_oracle_brute_force(orc, cornell, leaf):
    ctx = B.Context(-1)
    ctx.set_option("leaf_size", leaf)
    ctx.upload_scene(cornell["entities"], [m for _, m, _ in cornell["materials"]])
    st = ctx.stats()
    assert st["n_triangles"] == 17974 and 0 < st["bvh_depth"] <= 48
    S = orc.Scene(cornell["flat"])
    P = cornell["flat"]["positions"].reshape(-1, 3, 3)
    rng = np.random.default_rng(17 + leaf)
    hits = 0
    for i in range(1500):
        if i % 3 == 0:
            t = P[rng.integers(len(P))]
            o = (rng.dirichlet([1, 1, 1])[:, None] * t).sum(0)
        else:
            o = rng.uniform(-1.5, 1.5, 3) + [0, 1, 0]
        d = rng.normal(size=3)
        if i % 7 == 0:
            d[rng.integers(3)] = 0.0
        d /= np.linalg.norm(d)
        a = ctx.closest_hit_host(o, d)
        b = S.intersect(o, d, use_bvh=False)
        assert a == b
        hits += a[0]
    assert hits > 400


def test_bvh_depth_cap(orc, procedural, scene_io):
    # a long sliver strip forces SAH towards deep unbalanced trees; the builder must stay within the cap
    n = 3000
    x = np.arange(n + 1, dtype=np.float32) ** 2 * 1e-3
    v = np.stack([np.stack([x, 0 * x, 0 * x], 1), np.stack([x, 0 * x + 1e-3, 0 * x], 1)], 1).reshape(-1, 3)
    idx = np.array([[2 * i, 2 * i + 2, 2 * i + 1] for i in range(n)], np.int32)
    m = dict(vertices=v, normals=np.tile(np.float32([0, 0, 1]), (v.shape[0], 1)), texcoords=np.zeros((0, 2), np.float32), indices=idx)
    ctx = B.Context(-1)
    ctx.set_option("max_bvh_depth", 14)
    ctx.set_option("leaf_size", 2)
    ctx.upload_scene([(m, 0)], [scene_io.MAT_DEFAULT])
    assert ctx.stats()["bvh_depth"] <= 14
    flat = scene_io.flatten_scene([(m, 0)], [("a", scene_io.MAT_DEFAULT, "")])
    S = orc.Scene(flat)
    rng = np.random.default_rng(2)
    for _ in range(300):
        o = np.array([rng.uniform(0, x[-1]), rng.uniform(-1e-3, 2e-3), 1.0])
        d = np.array([rng.normal() * 0.01, rng.normal() * 0.001, -1.0])
        d /= np.linalg.norm(d)
        assert ctx.closest_hit_host(o, d) == S.intersect(o, d, use_bvh=False)


@pytest.mark.parametrize("W,H,tile,world", [(64, 48, 16, 2), (100, 37, 16, 3), (1920, 1080, 16, 8), (17, 9, 8, 4), (8, 8, 32, 1)])
def test_shard_pixels_partition(W, H, tile, world):
    parts = [B.shard_pixels(W, H, tile, r, world) for r in range(world)]
    allp = np.concatenate(parts)
    assert allp.size == W * H and np.array_equal(np.sort(allp), np.arange(W * H, dtype=np.uint32))
    if world > 1 and W * H >= 64 * world * 4:
        sizes = np.array([p.size for p in parts], np.float64)
        assert sizes.max() / sizes.min() < 1.6
    # the first 64 ids of a shard are one 8x8 screen block (one wave's starting patch)
    p = parts[0][:64]
    if W >= 8 and H >= 8:
        xs, ys = p % W, p // W
        assert xs.max() - xs.min() == 7 and ys.max() - ys.min() == 7


@pytest.mark.parametrize("leaf", [1, 4, 7])
def test_quad_nodes_cover_the_binary_tree(cornell, leaf):
    """The two-level collapse the wavefront kernel walks (PtNode4): every leaf reference of the binary tree sits in exactly one quad
    slot, all triangles are covered once, internal slots chain to exactly the other quad nodes, empty slots carry the never-hit box."""
    ctx = B.Context(-1)
    ctx.set_option("leaf_size", leaf)
    ctx.upload_scene(cornell["entities"], [m for _, m, _ in cornell["materials"]])
    q = ctx.quad_info()
    n_tris = cornell["flat"]["positions"].reshape(-1, 9).shape[0]
    assert q["triangles"] == n_tris
    assert q["leaf_slots"] == q["binary_leaf_refs"]
    assert q["internal_slots"] == q["quad_nodes"] - 1  # every quad node but the root is referenced once
    assert q["leaf_slots"] + q["internal_slots"] + q["empty_slots"] == 4 * q["quad_nodes"]
    assert 0 < q["quad_nodes"] < q["binary_nodes"] and 0 < q["depth"] <= (ctx.stats()["bvh_depth"] + 1) // 2 + 1
    ctx.close()


@pytest.mark.parametrize("leaf,wide", [(1, 1), (4, 1), (4, 0), (7, 1)])
def test_oct_nodes_cover_the_binary_tree(cornell, leaf, wide):
    """The three-level collapse the group walk reads (PtNode8, round 3): every triangle slot sits in exactly one leaf and inside that
    leaf's box (pt_debug_oct_info fails otherwise), every oct node but the root is referenced once, slots add up, and wide leaves
    (subtrees of <= 7 contiguous triangles as one leaf) never exceed the 3-bit count."""
    ctx = B.Context(-1)
    ctx.set_option("leaf_size", leaf)
    ctx.set_option("wide_leaves", wide)
    ctx.upload_scene(cornell["entities"], [m for _, m, _ in cornell["materials"]])
    o, q = ctx.oct_info(), ctx.quad_info()
    n_tris = cornell["flat"]["positions"].reshape(-1, 9).shape[0]
    assert o["triangles"] == n_tris == o["triangle_slots"]
    assert o["internal_slots"] == o["oct_nodes"] - 1
    assert o["leaf_slots"] + o["internal_slots"] + o["empty_slots"] == 8 * o["oct_nodes"]
    assert 0 < o["oct_nodes"] < q["quad_nodes"] and 0 < o["depth"] <= q["depth"]
    assert o["largest_leaf"] <= 7
    if wide:
        assert o["leaf_slots"] < q["binary_leaf_refs"] or leaf == 7  # leaves were merged
    else:
        assert o["leaf_slots"] == q["binary_leaf_refs"] and o["largest_leaf"] <= leaf
    ctx.close()


def test_rccl_load_failure_is_an_error_code_not_a_crash(tmp_path):
    """Round-2 advisor finding: a box without librccl crashed in the error path (dlerror() called twice, NULL into std::string).
    PT_RCCL_PATH forces the load to fail; the communicator calls must return PT_E_HIP with a message."""
    import subprocess, sys, os
    from conftest import ROOT

    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import ptamd; ptamd.load()\n"
        "from owl_path_tracer_amd.pyhost import binding as B\n"
        "import ctypes as C\n"
        "buf = (C.c_uint8 * 128)()\n"
        "rc = B.lib().pt_comm_get_unique_id(buf)\n"
        "msg = B.lib().pt_last_error(None).decode()\n"
        "assert rc == -5 or rc < 0, rc\n"
        "assert 'RCCL unavailable' in msg and 'no_such_rccl' in msg, msg\n"
        "rc2 = B.lib().pt_comm_get_unique_id(buf)\n"
        "assert rc2 == rc\n"
        "print('OK', rc, msg)\n" % ROOT
    )
    env = dict(os.environ, PT_RCCL_PATH=str(tmp_path / "no_such_rccl.so"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout


def _check_tiers(tiers, hist, capacity, ns):
    """Invariants of a tier table: consecutive queue ranges that cover every pixel, one tier per non-empty bucket in bucket order,
    workgroup ranges one after the other within the resident waves, pixels per wave from the plan's menu, waves that are needed."""
    menu = set(range(4, 17, 2)) | set(range(32, ns + 1, 8)) | {ns}
    q0 = w0 = 0
    assert [t["cost_class"] for t in tiers] == [b for b in range(32) if hist[b]]
    for t in tiers:
        assert t["q0"] == q0 and t["pixels"] == hist[t["cost_class"]] and t["wave0"] == w0
        assert t["per_wave"] in menu and t["waves"] >= 1
        assert (t["waves"] - 1) * t["per_wave"] < t["pixels"]  # no workgroup without a pixel
        q0 += t["pixels"]
        w0 += t["waves"]
    assert q0 == int(np.sum(hist)) and w0 <= capacity


def test_tier_plan_on_the_host():
    """pt_tiers.h - the plan a one-thread kernel makes on the device after the counting sort - run on the host (pt_debug_plan_tiers):
    a frame with a tail gets sparse waves for its expensive buckets and dense ones for the cheap majority, within the resident waves;
    a frame whose pixels all cost the same is left to the ring schedule unless forced; a very sparse launch always gets a plan; a
    launch with more pixels than slots fits by rounds."""
    cap, ns = 4096, 96
    # a shard of the dragon (1/8 of C4, roughly): 28 % sky, 45 % floor, the rest spread over 14 ever more expensive buckets
    hist = np.zeros(32, np.uint32)
    hist[31] = 72000; hist[26:29] = (21000, 33000, 25000); hist[19:26] = (7700, 9200, 10400, 9800, 8400, 8500, 10800)
    hist[10:19] = (50, 740, 2500, 3500, 4300, 6000, 8500, 8900, 8000)
    tiers = B.plan_tiers(hist, cap, ns)
    assert tiers, "a distribution with a tail must be planned"
    _check_tiers(tiers, hist, cap, ns)
    per = {t["cost_class"]: t["per_wave"] for t in tiers}
    assert per[10] <= 8 and per[31] == ns  # the longest chains in group-walk waves, the sky pixels in full ones
    assert all(per[a] <= per[b] for a, b in zip(sorted(per), sorted(per)[1:])), "cheaper buckets never get sparser waves"
    assert sum(t["waves"] for t in tiers) > 0.9 * cap  # the plan spends the waves it has
    # the same shape, eight times the pixels (the whole frame): does not fit one round - the plan uses rounds and still fits
    big = B.plan_tiers(hist * 8, cap, ns, force=True)
    _check_tiers(big, hist * 8, cap, ns)
    assert sum(t["waves"] * t["per_wave"] for t in big) < int(hist.sum()) * 8  # fewer slots than pixels: slots are reused
    # every pixel in two neighbouring buckets (a Cornell box): no tail -> the ring schedule (empty table) unless forced
    flat = np.zeros(32, np.uint32)
    flat[12:14] = (130000, 130000)
    assert B.plan_tiers(flat, cap, ns) == []
    forced = B.plan_tiers(flat, cap, ns, force=True)
    _check_tiers(forced, flat, cap, ns)
    # at most 16 pixels per resident wave: always planned, and nobody needs a dense wave
    sparse = np.zeros(32, np.uint32)
    sparse[12:14] = (16000, 16000)
    tiers = B.plan_tiers(sparse, cap, ns)
    _check_tiers(tiers, sparse, cap, ns)
    assert max(t["per_wave"] for t in tiers) <= 16
    # one pixel, one wave
    one = np.zeros(32, np.uint32)
    one[0] = 1
    tiers = B.plan_tiers(one, cap, ns)
    assert len(tiers) == 1 and tiers[0]["waves"] == 1 and tiers[0]["per_wave"] == 4
    # whatever the histogram and the number of resident waves: the plan fits or declines
    rng = np.random.default_rng(5)
    for _ in range(200):
        h = np.zeros(32, np.uint32)
        k = int(rng.integers(1, 12))
        h[rng.choice(32, k, replace=False)] = rng.integers(1, int(10 ** rng.uniform(0.5, 6.3)), k)
        cap_i = int(rng.choice([1, 2, 7, 64, 256, 2048, 4096]))
        ns_i = int(rng.choice([64, 96, 104]))
        t = B.plan_tiers(h, cap_i, ns_i, force=bool(rng.integers(0, 2)))
        if t:
            _check_tiers(t, h, cap_i, ns_i)
    # bad arguments are refused
    t = np.zeros(257, np.uint32)
    assert B.lib().pt_debug_plan_tiers(one.ctypes.data_as(C.POINTER(C.c_uint32)), 0, 96, 0, t.ctypes.data_as(C.POINTER(C.c_uint32)), 257) < 0
    assert B.lib().pt_debug_plan_tiers(one.ctypes.data_as(C.POINTER(C.c_uint32)), 4096, 96, 0, t.ctypes.data_as(C.POINTER(C.c_uint32)), 16) < 0


def test_render_kernel_instances_need_no_scratch():
    """pt_render refuses an instance of the wavefront kernel that spills to scratch (such builds rendered wrong pixels in round 1), so a
    source change that pushes the instrumented instance into scratch breaks every counted render and every scene without quad nodes -
    on the GPU box only.  hipcc reports the resource usage without a GPU: all five instances of both builds must show ScratchSize 0
    (found the hard way in round 4: a dynamic index into the counter block sent all of it to scratch)."""
    import re, shutil, subprocess, tempfile

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "owl-path-tracer_amd", "csrc")
    for extra in ([], ["-DPT_WITH_LOBE_BINS=1"]):
        with tempfile.TemporaryDirectory() as td:
            r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-S", "--cuda-device-only", "-o", os.path.join(td, "k.s"),
                                os.path.join(csrc, "pt_kernel.hip"), "-Rpass-analysis=kernel-resource-usage"] + extra, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        blocks = re.findall(r"Function Name: (\S*pt_render_wave_kernel\S*).*?ScratchSize \[bytes/lane\]: (\d+)", r.stderr, flags=re.S)
        assert len(blocks) == 5, blocks  # product / fallback, each with the fma and the subtracting slab form, + the instrumented instance
        assert all(int(sz) == 0 for _, sz in blocks), (extra, blocks)
