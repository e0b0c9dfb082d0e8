"""Randomised parity: many small random scenes / materials / cameras / schedules, product against the oracle, bit for bit.

The hand-written cases of test_gpu_parity.py test what somebody thought of; round 4's one real parity defect (a slab test that was wrong
for direction components of exactly 0) was found by brute force instead.  This file keeps brute force in the suite: every case draws a
triangle soup (with coincident, degenerate, axis-aligned and huge triangles mixed in), materials over the whole parameter range (the
extremes 0 / 1 over-represented), an environment mode, a camera (every fourth one axis-parallel from an off-origin point), an image size
that is not a multiple of anything, spp, depth and one of the library's render paths.  40 cases by default (~15 s); PT_FUZZ_CASES=N for
more, PT_FUZZ_SEED to move the sequence.  A failure prints the seed of the case, which reproduces it alone.
"""
import os

import numpy as np
import pytest

from owl_path_tracer_amd.pyhost import binding as B

pytestmark = pytest.mark.gpu  # (the scene generator below is also imported by the CPU property test in test_oracle_render.py)

N_CASES = int(os.environ.get("PT_FUZZ_CASES", "40"))
SEED0 = int(os.environ.get("PT_FUZZ_SEED", "20260405"))

PATHS = [(), (("groups", 2),), (("groups", 0),), (("kernel", 1),), (("whole", 1),), (("whole", 0), ("express_permille", 80)), (("schedule", 0), ("chunk_spp", 3)),
         (("fallback", 1),), (("box_exact", 1),), (("count", 1),), (("count", 1), ("quad", 0), ("groups", 0)), (("bvh", 1),), (("bvh", 2),), (("leaf", 1),), (("leaf", 7),),
         # scheduler knobs (none may change an image): launch geometry, pre-pass, cost filter, chunking, express waves, top-up threshold, retuning
         (("slots_per_wave", 64),), (("slots_per_wave", 104), ("blocks_per_cu", 2)), (("blocks_per_cu", 1),), (("prepass_spp", 1),), (("prepass_spp", 5), ("cost_radius", 0)),
         (("cost_radius", 7),), (("spp_per_launch", 5),), (("schedule", 0), ("chunk_spp", 1), ("chunk_tail_min", 0)), (("sticky_pct", 1),), (("sticky_pct", 100),),
         (("whole", 0), ("express_permille", 400), ("ns_express", 3)), (("tune0", 1),), (("tune0", 65),), (("adaptive", 0),), (("wide", 1),),
         (("shard", 0),), (("shard", 1),), (("shard", 2),)]
DEFAULTS = {"groups": 1, "kernel": 2, "whole": -1, "express_permille": -1, "schedule": 1, "chunk_spp": 64, "fallback": 0, "box_exact": -1, "count": 0, "quad": 1,
            "slots_per_wave": 0, "blocks_per_cu": 0, "prepass_spp": 0, "cost_radius": 2, "spp_per_launch": 0, "chunk_tail_min": -1, "sticky_pct": -1, "ns_express": 8,
            "tune0": 0, "adaptive": 1}


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _random_scene(rng):
    n_mesh = int(rng.integers(1, 5))
    scale = float(10.0 ** rng.uniform(-1.5, 2.0))  # scene sizes from centimetres to hundreds of units
    offset = rng.uniform(-1.0, 1.0, 3) * scale * (0.0 if rng.random() < 0.3 else float(rng.uniform(0.0, 3.0)))
    meshes = []
    for _ in range(n_mesh):
        kind = rng.integers(0, 5)
        n_tri = int(rng.integers(1, 160))
        if kind == 0:  # soup of small triangles
            c = rng.uniform(-1, 1, (n_tri, 1, 3))
            v = c + rng.normal(0, 0.25, (n_tri, 3, 3))
        elif kind == 1:  # axis-aligned rectangles (two triangles each), some coincident
            v = []
            for _k in range((n_tri + 1) // 2):
                ax = int(rng.integers(0, 3))
                lo, hi = np.sort(rng.uniform(-1, 1, (2, 3)), axis=0)
                lo[ax] = hi[ax] = float(rng.choice([-1.0, 0.0, 0.5, rng.uniform(-1, 1)]))
                a, b = (ax + 1) % 3, (ax + 2) % 3
                p = [lo.copy(), lo.copy(), hi.copy(), lo.copy()]
                p[1][a] = hi[a]
                p[3][b] = hi[b]
                v += [[p[0], p[1], p[2]], [p[0], p[2], p[3]]]
                if rng.random() < 0.2:
                    v += [[p[0], p[1], p[2]]]  # a coincident copy
            v = np.asarray(v)
        elif kind == 2:  # a few huge triangles
            v = rng.uniform(-6, 6, (min(n_tri, 4), 3, 3))
        elif kind == 3:  # degenerate: repeated vertices, needles
            v = rng.uniform(-1, 1, (n_tri, 3, 3))
            v[::2, 1] = v[::2, 0]
            v[1::3, 2] = v[1::3, 1] + 1e-7
        else:  # a closed box (rays bounce many times inside)
            lo, hi = np.array([-1.2, -1.2, -1.2]), np.array([1.2, 1.2, 1.2])
            c = np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])])
            f = [[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]]
            v = c[np.asarray(f)]
        v = (np.asarray(v, np.float64) * scale + offset).astype(np.float32).reshape(-1, 3)
        nrm = rng.normal(0, 1, v.shape)
        if rng.random() < 0.5:  # geometric normals for half of the meshes
            t = v.reshape(-1, 3, 3).astype(np.float64)
            g = np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0])
            g[np.linalg.norm(g, axis=1) < 1e-20] = (0, 1, 0)
            nrm = np.repeat(g, 3, axis=0)
        nrm = nrm / np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
        meshes.append(dict(vertices=v, normals=nrm.astype(np.float32), texcoords=rng.uniform(-2, 3, (len(v), 2)).astype(np.float32),
                           indices=np.arange(len(v), dtype=np.int32).reshape(-1, 3)))
    n_mat = int(rng.integers(1, 5))
    mats = np.zeros((n_mat, 17), np.float32)
    for m in mats:
        def pick():
            r = rng.random()
            return 0.0 if r < 0.2 else (1.0 if r < 0.4 else float(rng.random()))
        m[0:3] = [pick() for _ in range(3)]
        for k in range(3, 13):
            m[k] = pick()
        m[13] = float(rng.choice([1.0, 1.45, 1.5, 2.4, rng.uniform(1.0, 3.0)]))  # ior
        m[14] = pick() if rng.random() < 0.5 else 0.0
        m[15] = pick()
        m[16] = float(rng.choice([0.0, 0.0, 0.0, rng.uniform(0.5, 20.0)]))  # emission
    ents = [(mesh, int(rng.integers(0, n_mat))) for mesh in meshes]
    return ents, mats, scale, offset


def _random_camera(rng, scale, offset, W, H):
    centre = offset + rng.uniform(-0.3, 0.3, 3) * scale
    if rng.random() < 0.25:  # exactly axis-parallel, from a point whose coordinates are not 0
        ax = int(rng.integers(0, 3))
        frm = centre.copy()
        frm[ax] += float(rng.choice([-1.0, 1.0])) * scale * float(rng.uniform(0.5, 4.0))
        up = [0.0, 0.0, 0.0]
        up[(ax + 1 + int(rng.integers(0, 2))) % 3] = 1.0
        fov = float(rng.choice([0.5, 5.0, 40.0, 90.0]))
    else:
        frm = centre + rng.normal(0, 1, 3) * scale * float(rng.uniform(0.3, 2.5))
        up = list(rng.normal(0, 1, 3))
        fov = float(rng.uniform(5.0, 110.0))
    return [float(x) for x in frm], [float(x) for x in centre], [float(x) for x in up], fov


def draw_case(seed, large=False):
    """Everything one case consists of, drawn from `seed` in a fixed order (tools/fuzz_bisect.py and tools/fuzz_diag.py use it too)."""
    rng = np.random.default_rng(seed)
    ents, mats, scale, offset = _random_scene(rng)
    W, H = int(rng.integers(1, 90)), int(rng.integers(1, 70))
    spp = int(rng.choice([1, 2, 7, 16, 33, 64, 130]))
    if large:  # more pixels than the chip has path slots (393 216): tickets, rings, laggards, tiers with rounds, express waves at scale
        W, H = int(rng.integers(500, 1300)), int(rng.integers(400, 900))
        spp = int(rng.choice([4, 9, 20, 33, 48]))
    depth = int(rng.choice([1, 2, 4, 16, 31]))
    mode = int(rng.integers(0, 3))
    texs, mesh_tex, tex_by_mat = None, None, None
    if rng.random() < 0.3:  # textures on the meshes of material 0
        h, w = int(rng.integers(1, 9)), int(rng.integers(1, 9))
        px = rng.integers(0, 256, (h, w, 3)).astype(np.uint32)
        t = (px[..., 0] | (px[..., 1] << 8) | (px[..., 2] << 16) | (0xFF << 24)).astype(np.uint32)
        texs, mesh_tex, tex_by_mat = [t], [0 if mid == 0 else -1 for _, mid in ents], {0: t}
    if mode == 0:
        env = dict(use_auto=True, intensity=float(rng.uniform(0.0, 2.0)))
    elif mode == 1:
        env = dict(color=tuple(float(x) for x in rng.random(3)), intensity=float(rng.choice([0.0, 1.0, rng.uniform(0, 3)])))
    else:
        eh, ew = int(rng.integers(1, 17)), int(rng.integers(1, 33))
        px = rng.integers(0, 256, (eh, ew, 3)).astype(np.uint32)
        env = dict(use_map=True, intensity=float(rng.uniform(0.2, 2.0)), env_map=(px[..., 0] | (px[..., 1] << 8) | (px[..., 2] << 16) | (0xFF << 24)).astype(np.uint32))
    frm, at, up, fov = _random_camera(rng, scale, offset, W, H)
    path = PATHS[int(rng.integers(0, len(PATHS)))]
    if large and path and path[0][0] in ("kernel", "count"):
        path = (("schedule", 0), ("chunk_spp", 2))  # (the slow instances are not what this variant is about)
    shard = None
    for k, v in path:
        if k == "shard":  # one rank's pixel tiles of a multi-GPU frame: exactly the full frame's values there, zero elsewhere
            world = int(rng.integers(2, 10))
            shard = (int(rng.integers(0, world)), world, int(rng.choice([1, 4, 16, 32])))
    return dict(seed=seed, ents=ents, mats=mats, W=W, H=H, spp=spp, depth=depth, mode=mode, env=env, texs=texs, mesh_tex=mesh_tex, tex_by_mat=tex_by_mat,
                camera=(frm, at, up, fov), path=path, shard=shard)


def _case(gpu, orc, scene_io, seed, large=False, path=None):
    c = draw_case(seed, large)
    ents, mats, W, H, spp, depth, mode, env = c["ents"], c["mats"], c["W"], c["H"], c["spp"], c["depth"], c["mode"], c["env"]
    texs, mesh_tex, tex_by_mat, shard = c["texs"], c["mesh_tex"], c["tex_by_mat"], c["shard"]
    frm, at, up, fov = c["camera"]
    if path is None:
        path = c["path"]
    else:
        shard = None  # (tools/fuzz_diag.py: the same case through a path of the caller's choice)
    cam = B.to_camera_data(frm, at, up, fov, W, H)
    ocam = orc.to_camera_data(tuple(frm), tuple(at), tuple(up), fov, W, H)
    pre = {"bvh": ("bvh_builder", 3), "leaf": ("leaf_size", 4), "wide": ("wide_leaves", 0)}
    try:
        for k, v in path:
            if k in pre:
                gpu.set_option(pre[k][0], v)
        gpu.upload_scene(ents, mats, textures=texs, mesh_textures=mesh_tex, env=B.make_env(**env))
        for k, v in path:
            if k not in pre and k != "shard":
                gpu.set_option(k, v)
        if shard:
            gpu.set_pixel_shard(*shard)
        got, got8 = gpu.render(cam, W, H, spp, depth, want_rgba8=True)
    finally:
        gpu.set_pixel_shard(0, 1, 16)
        for k, v in path:
            if k in pre:
                gpu.set_option(*pre[k])
            elif k != "shard":
                gpu.set_option(k, DEFAULTS[k])
    S = orc.Scene(scene_io.flatten_scene(ents, [("m%d" % i, m, "") for i, m in enumerate(mats)], tex_by_mat))
    want, want8, _ = S.render(ocam, orc.make_env(**env), W, H, spp, depth, want_rgba8=True)
    if shard:
        own = np.zeros(W * H, bool)
        own[B.shard_pixels(W, H, shard[2], shard[0], shard[1])] = True
        own = own.reshape(H, W)[::-1]
        want = np.where(own[..., None], want, np.float32(0.0))
        want8 = np.where(own, want8, np.uint32(0))
    assert (got8 == want8).all(), "fuzz case seed=%d: RGBA8 output differs (path %s)" % (seed, path)
    same = (_bits(got) == _bits(want)) | (np.isnan(got) & np.isnan(want))
    if not same.all():
        bad = np.argwhere(~same)
        raise AssertionError("fuzz case seed=%d (%dx%d, %d spp, depth %d, env mode %d, path %s, %d triangles): %d of %d values differ; first at %s: gpu=%r oracle=%r"
                             % (seed, W, H, spp, depth, mode, path, sum(len(m["indices"]) for m, _ in ents), len(bad), same.size, bad[0], got[tuple(bad[0])], want[tuple(bad[0])]))


def _context():
    gpu = B.Context(0)
    for kv in os.environ.get("PT_FUZZ_OPTIONS", "").split():  # e.g. PT_LIB_PATH=.../libmi355pt_lobebins.so PT_FUZZ_OPTIONS="lobe_bins=1": a side build through the same cases
        k, v = kv.split("=")
        gpu.set_option(k, int(v))
    return gpu


def test_random_scenes_bitwise(orc, scene_io):
    gpu = _context()
    only = os.environ.get("PT_FUZZ_ONLY")
    seeds = [int(only)] if only else [SEED0 + i for i in range(N_CASES)]
    for n, seed in enumerate(seeds):
        _case(gpu, orc, scene_io, seed)
        if (n + 1) % 100 == 0:
            print("fuzz: %d cases bit-identical" % (n + 1), flush=True)


N_LARGE = int(os.environ.get("PT_FUZZ_LARGE", "8"))


def test_random_large_frames_bitwise(orc, scene_io):
    """The same generator at 200 000 - 1 100 000 pixels and 4-48 spp: most of these frames have more pixels than the chip has path slots, so the ticket
    counters, the per-chunk rings with their cross-wave hand-offs, the tier plan with rounds and the express waves all run at scale -
    with scenes whose pixels differ wildly in cost.  8 cases by default (~10 s); PT_FUZZ_LARGE=N."""
    gpu = _context()
    for n in range(N_LARGE):
        _case(gpu, orc, scene_io, SEED0 + 5_000_000 + n, large=True)
        if (n + 1) % 25 == 0:
            print("fuzz (large frames): %d cases bit-identical" % (n + 1), flush=True)
