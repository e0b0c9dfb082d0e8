"""Oracle render-loop checks: BVH == brute force, scene ingestion quirks, and the trace_path / ray_gen
quirks of SURVEY.md appendix B (device.cu:113-254)."""
import os

import numpy as np
import pytest


def _cam(orc, c, W, H):
    return orc.to_camera_data(c["look_from"], c["look_at"], c["look_up"], c["vertical_fov"], W, H)


def test_scene_ingestion_cornell(cornell):
    # SURVEY appendix A; mesh_loader.cpp:9-83 re-indexes by position index only
    got = [(n, m["indices"].shape[0]) for n, m in cornell["meshes"]]
    assert got == [("box", 10), ("wall_left", 2), ("wall_right", 2), ("wall_tbb", 6), ("sphere", 17952), ("light", 2)]
    assert [n for n, _, _ in cornell["materials"]] == ["box", "sphere", "light", "wall_left", "wall_right", "wall_tbb"]
    # entity material id = index in JSON order (application.cpp:166-179)
    assert [mid for _, mid in cornell["entities"]] == [0, 3, 4, 5, 1, 2]
    assert cornell["flat"]["positions"].shape == (17974, 9)
    sph = dict(cornell["meshes"])["sphere"]
    assert sph["vertices"].shape[0] == sph["normals"].shape[0] == sph["texcoords"].shape[0]
    assert sph["vertices"].shape[0] < 3 * 17952  # shared vertices -> smooth ("first seen") normals
    assert cornell["materials"][2][1][16] == 15.0  # light emission


def test_scene_ingestion_cube(cube):
    (name, m), = cube["meshes"]
    assert name == "cube" and m["indices"].shape == (12, 3) and m["vertices"].shape == (36, 3)
    assert cube["materials"][0][2] == "cube-textures/cube.png"  # parser.cpp:34
    assert cube["flat"]["texture_index"].tolist() == [0] * 12


def test_create_mesh_first_seen_normals(scene_io, tmp_path):
    p = tmp_path / "t.obj.scene"
    p.write_text("o a\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nvn 0 0 1\nvn 0 1 0\nf 1//1 2//1 3//1\nf 2//2 4//2 3//2\no unused\nv 0 0 1\nf 1 2 5\n")
    meshes = scene_io.load_obj(str(p))
    assert [n for n, _ in meshes] == ["a", "unused"]
    a = meshes[0][1]
    assert a["indices"].tolist() == [[0, 1, 2], [1, 3, 2]]
    # vertices 1 and 2 keep the normal of the first corner that introduced them; vertex 3 gets the second face's
    assert a["normals"].tolist() == [[0, 0, 1], [0, 0, 1], [0, 0, 1], [0, 1, 0]]
    # objects without a same-named material vanish (quirk 21)
    ents = scene_io.build_entities(meshes, [("a", scene_io.MAT_DEFAULT, "")])
    assert len(ents) == 1 and ents[0][1] == 0


def test_bvh_equals_brute_force(orc, cornell, cube):
    rng = np.random.default_rng(11)
    for sc, n_rays in ((cube, 3000), (cornell, 1500)):
        S = orc.Scene(sc["flat"])
        P = sc["flat"]["positions"].reshape(-1, 3, 3)
        lo, hi = P.min((0, 1)), P.max((0, 1))
        hits = 0
        for i in range(n_rays):
            if i % 3 == 0:  # secondary-like: origin ON a surface point
                t = P[rng.integers(len(P))]
                b = rng.dirichlet([1, 1, 1])
                o = (b[:, None] * t).sum(0)
            else:
                o = rng.uniform(lo - 1, hi + 1)
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
            if i % 7 == 0:
                d[rng.integers(3)] = 0.0  # axis-parallel component (inf in the slab test)
                d /= np.linalg.norm(d)
            a = S.intersect(o, d, use_bvh=True)
            b_ = S.intersect(o, d, use_bvh=False)
            assert a == b_
            hits += a[0]
        assert hits > n_rays // 4


def test_bvh_leaf_size_independent(orc, cornell):
    S1, S2 = orc.Scene(cornell["flat"], leaf_size=1), orc.Scene(cornell["flat"], leaf_size=7)
    rng = np.random.default_rng(5)
    for _ in range(500):
        o = rng.uniform(-1, 1, 3) + [0, 1, 0]
        d = rng.normal(size=3)
        assert S1.intersect(o, d / np.linalg.norm(d)) == S2.intersect(o, d / np.linalg.norm(d))


def test_camera(orc):
    # camera.cpp:3-21
    cam = orc.to_camera_data([3, 1, 0], [0, 1, 0], [0, 1, 0], 50, 512, 256).as_array()
    h = np.tan(np.radians(50) / 2)
    np.testing.assert_allclose(cam[0:3], [3, 1, 0])
    np.testing.assert_allclose(cam[6:9], [0, 0, -2 * h * 2], atol=1e-6)  # u = up x w = (0,0,-1); width = aspect*2h
    np.testing.assert_allclose(cam[9:12], [0, 2 * h, 0], atol=1e-6)
    np.testing.assert_allclose(cam[3:6], [3 - 1, 1 - h, 2 * h], atol=1e-6)


def test_render_quirks_miss_and_emission(orc, scene_io, procedural):
    W = H = 16
    cam = orc.to_camera_data([0, 1, 5], [0, 1, 0], [0, 1, 0], 40, W, H)
    # a big emitter in front of the camera: radiance = vec3(emission) (assignment, white, two-sided; device.cu:157-161)
    q = procedural.quad((-50, -50, 0), (50, -50, 0), (50, 50, 0), (-50, 50, 0), (0, 0, -1))  # faces AWAY
    flat = scene_io.flatten_scene([(q, 0)], [("e", scene_io.material(emission=3.5, base_color=[1, 0, 0]), "")])
    S = orc.Scene(flat)
    rgb, rgba, _ = S.render(cam, orc.make_env(color=(9, 9, 9), intensity=1), W, H, 4, 8, want_rgba8=True)
    np.testing.assert_array_equal(rgb, np.full((H, W, 3), 3.5, np.float32))
    assert (rgba == 0xFFFFFFFF).all()  # linear 8-bit, clamped (no gamma; device.cu:248,252)
    # empty scene: miss -> (0 + env_color) * intensity (device.cu:136-148)
    E = orc.Scene(scene_io.flatten_scene([], [("e", scene_io.MAT_DEFAULT, "")]))
    rgb, _, _ = E.render(cam, orc.make_env(color=(0.25, 0.5, 1.0), intensity=0.5), W, H, 3, 8)
    np.testing.assert_allclose(rgb, np.broadcast_to(np.float32([0.125, 0.25, 0.5]), (H, W, 3)), rtol=1e-6)
    # environment_auto sky: lerp(1, (.5,.7,1), .5*(d.y+1)) and row flip (device.cu:141,251): top rows look up -> bluer
    rgb, _, _ = E.render(cam, orc.make_env(use_auto=True, intensity=1), W, H, 8, 8)
    assert rgb[0, :, 0].mean() < rgb[-1, :, 0].mean()
    assert np.all(rgb[..., 2] == 1.0)
    # intensity 0 kills misses
    rgb, _, _ = E.render(cam, orc.make_env(use_auto=True, intensity=0), W, H, 2, 8)
    assert not rgb.any()


def test_render_depth_limit_and_counters(orc, cube):
    S = orc.Scene(cube["flat"])
    W = H = 24
    cam = _cam(orc, cube["camera"], W, H)
    env = orc.make_env(use_auto=True, intensity=1)
    rgb, _, cnt = S.render(cam, env, W, H, 8, 4, want_counters=True)
    assert cnt["samples"] == W * H * 8
    assert cnt["samples"] <= cnt["rays"] <= 4 * cnt["samples"] + cnt["nan_retries"]
    assert np.isfinite(rgb).all() and rgb.max() <= 1.0 + 1e-6
    # max_depth 1: a hit contributes nothing unless emissive; only misses carry radiance
    rgb1, _, cnt1 = S.render(cam, env, W, H, 8, 1, want_counters=True)
    assert cnt1["rays"] == cnt1["samples"]
    assert (rgb1 <= rgb + 1e-6).all()


def test_render_deterministic_threads_and_pixel_list(orc, cube):
    S = orc.Scene(cube["flat"])
    W, H = 20, 12
    cam = _cam(orc, cube["camera"], W, H)
    env = orc.make_env(use_auto=True, intensity=1)
    a, _, _ = S.render(cam, env, W, H, 6, 4, threads=1)
    b, _, _ = S.render(cam, env, W, H, 6, 4, threads=5)
    c, _, _ = S.render(cam, env, W, H, 6, 4, use_bvh=False)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(a, c)
    # disjoint pixel subsets sum (with zeros) to the full image bit-for-bit (basis of the multi-GPU reduce)
    ids = np.arange(W * H, dtype=np.uint32)
    p0, _, _ = S.render(cam, env, W, H, 6, 4, pixel_list=ids[0::2])
    p1, _, _ = S.render(cam, env, W, H, 6, 4, pixel_list=ids[1::2])
    np.testing.assert_array_equal(p0 + p1, a)
    # per-pixel trace: stream is sequential in samples (device.cu:226-243)
    rgbs, states = S.trace_pixel(cam, env, W, H, 7, 5, 6, 4)
    np.testing.assert_allclose(rgbs.sum(0) / 6, a[H - 1 - 5, 7], rtol=1e-6)
    rgbs3, states3 = S.trace_pixel(cam, env, W, H, 7, 5, 3, 4)
    np.testing.assert_array_equal(states3, states[:3])


def test_furnace_known_answer(orc, scene_io, procedural):
    # thesis furnace anchor: white environment, diffuse sphere (assets/sphere.json material,
    # base 1, roughness 1) -> the Disney diffuse lobe is >= Lambert, image ~>= 1 on the sphere
    meshes = procedural.furnace_sphere(24)
    mat = scene_io.material(base_color=[1, 1, 1], specular=0.0, specular_tint=0.0, roughness=1.0, sheen_tint=0.0, clearcoat_gloss=0.0, ior=1.5)
    flat = scene_io.flatten_scene(scene_io.build_entities(meshes, [("sphere", mat, "")]), [("sphere", mat, "")])
    S = orc.Scene(flat)
    W = H = 32
    cam = orc.to_camera_data([3, 1, 0], [0, 1, 0], [0, 1, 0], 50, W, H)
    rgb, rgba, _ = S.render(cam, orc.make_env(color=(1, 1, 1), intensity=1), W, H, 64, 16, want_rgba8=True)
    centre = rgb[12:20, 12:20]
    assert 0.97 < centre.mean() < 1.25
    assert ((rgba[12:20, 12:20] & 0xFF) >= 240).all()


def test_golden_regression(orc, cube):
    """Self-generated regression pin (tests/golden/make_golden.py) -- NOT a reference fixture."""
    from conftest import GOLDEN

    path = os.path.join(GOLDEN, "cube_32x32_8spp_d4_oracle.npy")
    S = orc.Scene(cube["flat"])
    cam = _cam(orc, cube["camera"], 32, 32)
    rgb, _, _ = S.render(cam, orc.make_env(use_auto=True, intensity=1), 32, 32, 8, 4)
    np.testing.assert_array_equal(rgb, np.load(path))


@pytest.mark.parametrize("key", ["diffuse_roughness(0.0)", "diffuse_roughness(1.0)", "metallic_roughness(0.0)", "specular_transmission_roughness(0.0)", "metallic_vndf_roughness(1.0)"])
def test_furnace_against_reference_rendered_images(orc, scene_io, procedural, key):
    """The oracle against the ONLY rendered outputs the reference repository holds (thesis/assets/furnace-test): radial profile of
    the 8-bit image, ring by ring (tests/furnace_common.py).  This pins, with data produced by the reference itself: the camera, the
    miss shader, the diffuse lobe (sampling, eval, pdf), the mirror limit of the specular and glass lobes, `f |cos| / pdf`
    accumulation and the truncating x256 quantiser assumed for owl::make_rgba (0.976 -> a 249/250 mix, mean 249.4, as in the PNG)."""
    import furnace_common as F

    ents, mats = F.setup(scene_io, procedural, key)
    S = orc.Scene(scene_io.flatten_scene(ents, mats))
    cam = orc.to_camera_data([3, 1, 0], [0, 1, 0], [0, 1, 0], 50, F.W, F.H)
    _, rgba, _ = S.render(cam, orc.make_env(color=(1, 1, 1), intensity=1), F.W, F.H, F.SPP, F.DEPTH, want_rgba8=True)
    F.check(key, rgba)


def test_c4_golden_crc_is_the_oracles():
    """The checksum bench.py asserts after its timed loop (tests/golden/c4_frame_crc.json) was first written by bench.py itself; since round 4
    it is tied to the oracle: the whole C4 frame at full spp rendered by the ORACLE on the GPU box's host cores has the same crc32
    (recorded by the opt-in GPU test test_whole_frames_at_full_spp in profiles/r04_full_frame_parity.json)."""
    import json
    root = os.path.join(os.path.dirname(__file__), "..")
    golden = json.load(open(os.path.join(root, "tests", "golden", "c4_frame_crc.json")))["crc32_float3_frame"]
    recs = json.load(open(os.path.join(root, "profiles", "r04_full_frame_parity.json")))
    c4 = [r for r in recs if r["config"].startswith("C4 ")]
    assert len(c4) == 1 and c4[0]["pixels_compared"] == 1920 * 1080 and c4[0]["spp"] == 1024
    assert c4[0]["bit_identical"] and c4[0]["crc32_oracle"] == golden == c4[0]["crc32_gpu"]
    assert all(r["bit_identical"] and all(v["bit_identical"] for v in r.get("variants", [])) for r in recs)


def test_closest_hit_does_not_depend_on_the_hierarchy(orc, scene_io):
    """The closest hit is DEFINED as the minimum over all triangles of the Moeller-Trumbore t (ties: lower id), so that every hierarchy -
    the oracle's, the product's three builders, quad and oct nodes - must find the same one.  That only holds if no triangle reports hits
    outside its (padded) bounding box: slivers, whose test results are rounding noise, are collapsed at scene build (pt_oracle.c
    orc_scene_create, csrc/pt_api.cpp pt_collapse_sliver).  Property test on the oracle alone: BVH walk == brute force over random
    scenes with needles, coincident and degenerate triangles (the generator of tests/test_gpu_fuzz.py)."""
    import test_gpu_fuzz as F

    n_rays = 0
    for seed in range(F.SEED0, F.SEED0 + 150):
        rng = np.random.default_rng(seed)
        ents, mats, scale, offset = F._random_scene(rng)
        S = orc.Scene(scene_io.flatten_scene(ents, [("m%d" % i, m, "") for i, m in enumerate(mats)], None))
        P = np.concatenate([m["vertices"][m["indices"]].reshape(-1, 3) for m, _ in ents])
        for k in range(120):
            if k % 2:
                o = P[rng.integers(len(P))] + rng.normal(0, 1e-3, 3) * scale  # origins on / near the geometry, like secondary rays
            else:
                o = offset + rng.normal(0, 1, 3) * scale * 2.0
            d = rng.normal(0, 1, 3)
            if k % 7 == 0:
                d[rng.integers(3)] = 0.0
            d = (d / np.linalg.norm(d)).astype(np.float32)
            a = S.intersect(np.float32(o), d, use_bvh=True)
            b = S.intersect(np.float32(o), d, use_bvh=False)
            assert a[0] == b[0] and (not a[0] or (np.float32(a[1:4]).view(np.uint32) == np.float32(b[1:4]).view(np.uint32)).all() and a[4] == b[4]), (seed, k, a, b)
            n_rays += 1
    assert n_rays == 150 * 120


def test_oracle_image_is_the_same_with_and_without_its_hierarchy(orc, scene_io):
    """Whole images of random scenes (tests/test_gpu_fuzz.py's generator) rendered by the oracle through its BVH and by brute force over all
    triangles: bit-identical - the image-level form of the order-independence the parity claim rests on - and the same on 1 and 4 threads and
    through the pixel-list entry point."""
    import test_gpu_fuzz as F

    for seed in range(F.SEED0 + 100, F.SEED0 + 130):
        rng = np.random.default_rng(seed)
        ents, mats, scale, offset = F._random_scene(rng)
        W, H = int(rng.integers(4, 40)), int(rng.integers(4, 30))
        S = orc.Scene(scene_io.flatten_scene(ents, [("m%d" % i, m, "") for i, m in enumerate(mats)], None))
        frm, at, up, fov = F._random_camera(rng, scale, offset, W, H)
        cam = orc.to_camera_data(tuple(frm), tuple(at), tuple(up), fov, W, H)
        env = orc.make_env(use_auto=True, intensity=1.0)
        a, _, _ = S.render(cam, env, W, H, 12, 8, use_bvh=True, threads=4)
        b, _, _ = S.render(cam, env, W, H, 12, 8, use_bvh=False, threads=1)
        same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
        assert same.all(), (seed, np.argwhere(~same)[:3])
        ids = np.arange(W * H, dtype=np.uint32)[::3]
        sub = np.zeros((H, W, 3), np.float32)
        S.render(cam, env, W, H, 12, 8, pixel_list=ids, out=sub)
        ys, xs = (H - 1 - ids // W), ids % W
        s2 = (sub[ys, xs].view(np.uint32) == a[ys, xs].view(np.uint32)) | (np.isnan(sub[ys, xs]) & np.isnan(a[ys, xs]))
        assert s2.all(), seed
