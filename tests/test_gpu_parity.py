"""GPU parity tests (run on the MI355X box with -m gpu).  Every check goes through the C-ABI (libmi355pt.so) and
compares with the CPU oracle on the same seeded inputs.

Tolerance: the kernel and the oracle implement one arithmetic contract (IEEE binary32, explicit fma only, correctly
rounded div/sqrt, deterministic polynomial transcendentals), so the bar is BIT-EXACT equality of the float framebuffer
-- per-pixel L2 == 0 against the oracle.  (Against the reference's OptiX build nothing can be measured here; see DESIGN.md.)
"""
import os

import numpy as np
import pytest

from owl_path_tracer_amd.pyhost import binding as B

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    ctx = B.Context(0)
    yield ctx
    ctx.close()


def _host_threads():
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_bitwise(a, b, what=""):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    same = (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))
    if not same.all():
        bad = np.argwhere(~same)
        raise AssertionError("%s: %d of %d values differ; first at %s: gpu=%r oracle=%r" % (what, len(bad), same.size, bad[0], a[tuple(bad[0])], b[tuple(bad[0])]))


def _mats(sc):
    return [m for _, m, _ in sc["materials"]]


def _upload(ctx, sc, textures=None, mesh_textures=None, env=None):
    ctx.upload_scene(sc["entities"], _mats(sc), textures=textures, mesh_textures=mesh_textures, env=env)


# The oracle gets ITS OWN camera (camera.cpp:3-21 restated in oracle/pt_oracle.c) from the look-at parameters, not the 48 bytes the product's
# pt_to_camera_data computed: a deviation in either shows as a parity failure instead of cancelling out (round-3 verdict, weak #2).
_CAM_ARGS = {}


def mkcam(look_from, look_at, look_up, vfov, W, H):
    cam = B.to_camera_data(look_from, look_at, look_up, vfov, W, H)
    _CAM_ARGS[cam.as_array().tobytes()] = (tuple(look_from), tuple(look_at), tuple(look_up), float(vfov), int(W), int(H))
    return cam


def _cam(sc, W, H):
    c = sc["camera"]
    return mkcam(c["look_from"], c["look_at"], c["look_up"], c["vertical_fov"], W, H)


def _ocam(orc, cam):
    args = _CAM_ARGS[cam.as_array().tobytes()]  # KeyError: build the camera with mkcam()
    return orc.to_camera_data(*args)


# ---------------------------------------------------------------------------------------------------------------------
# building blocks
# ---------------------------------------------------------------------------------------------------------------------

def test_deterministic_libm_bitwise(gpu, orc):
    rng = np.random.default_rng(99)
    n = 200_000
    x = rng.uniform(-8, 8, n).astype(np.float32)
    for fn in ("sin", "cos", "tan"):
        assert_bitwise(gpu.debug_eval(fn, x, 1)[:, 0], orc.dm(fn, x), fn)
    x = np.concatenate([rng.uniform(-50, 50, n // 2), rng.standard_cauchy(n // 2) * 100, [0, np.inf, -np.inf, 1e-30]]).astype(np.float32)
    assert_bitwise(gpu.debug_eval("atan", x, 1)[:, 0], orc.dm("atan", x), "atan")
    x = rng.uniform(-1, 1, n).astype(np.float32)
    assert_bitwise(gpu.debug_eval("asin", x, 1)[:, 0], orc.dm("asin", x), "asin")
    x = np.exp(rng.uniform(-30, 30, n)).astype(np.float32)
    assert_bitwise(gpu.debug_eval("log", x, 1)[:, 0], orc.dm("log", x), "log")
    x = rng.uniform(-90, 90, n).astype(np.float32)
    assert_bitwise(gpu.debug_eval("exp", x, 1)[:, 0], orc.dm("exp", x), "exp")
    xy = np.stack([rng.uniform(0, 1, n), rng.uniform(0, 3, n)], 1).astype(np.float32)
    xy[:100, 1] = 2.2
    xy[100:110, 0] = [0, 1, 0.5, 2, 4, 1e-7, 1e-20, 0.25, 0.75, 1e-3]
    assert_bitwise(gpu.debug_eval("pow", xy, 1)[:, 0], orc.dm("pow", xy[:, 0], xy[:, 1]), "pow")
    yx = rng.uniform(-3, 3, (n, 2)).astype(np.float32)
    yx[:50, 1] = 0
    assert_bitwise(gpu.debug_eval("atan2", yx, 1)[:, 0], orc.dm("atan2", yx[:, 0], yx[:, 1]), "atan2")
    # division and sqrt must be the correctly rounded IEEE operations on the GPU
    x = np.exp(rng.uniform(-40, 40, n)).astype(np.float32)
    y = rng.uniform(-3, 3, n).astype(np.float32)
    assert_bitwise(gpu.debug_eval("sqrt", x, 1)[:, 0], np.sqrt(x), "sqrt")
    assert_bitwise(gpu.debug_eval("div", np.stack([x, y], 1), 1)[:, 0], x / y, "div")


@pytest.mark.skipif(os.environ.get("PT_LIBM_EXHAUSTIVE") != "1", reason="opt-in (PT_LIBM_EXHAUSTIVE=1): every float32 through every unary function, ~3 minutes")
def test_deterministic_libm_exhaustive(gpu, orc):
    """ALL 2^32 float32 bit patterns through sin, cos, tan, atan, asin, log, exp and sqrt on the GPU (pt_device.h) and in the oracle
    (pt_oracle.c): identical bits everywhere (any NaN == any NaN; sin / cos / tan: everywhere below |x| = 2e9).  The two-argument functions (atan2, pow, division) get the cross
    product of 4 096 special and random values (16.7 M pairs each).  Record: gpurun_out/libm_exhaustive.json."""
    import json, time
    from concurrent.futures import ThreadPoolExecutor

    threads = _host_threads()
    rec = {}
    chunk = 1 << 26

    def oracle_eval(fn, x, y=None):
        parts = np.array_split(np.arange(x.size), threads * 4)
        out = np.empty_like(x)

        def work(idx):
            out[idx[0]:idx[-1] + 1] = orc.dm(fn, x[idx[0]:idx[-1] + 1], None if y is None else y[idx[0]:idx[-1] + 1])
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(work, [p for p in parts if p.size]))
        return out

    for fn in ("sin", "cos", "tan", "atan", "asin", "log", "exp", "sqrt"):
        t0 = time.perf_counter()
        bad, bad_inside, smallest = 0, 0, float("inf")
        for base in range(0, 1 << 32, chunk):
            x = np.arange(base, base + chunk, dtype=np.uint64).astype(np.uint32).view(np.float32)
            g = gpu.debug_eval(fn, x, 1)[:, 0]
            w = oracle_eval(fn, x)
            same = (g.view(np.uint32) == w.view(np.uint32)) | (np.isnan(g) & np.isnan(w))
            if not same.all():
                i = int(np.argmin(same))
                bad += int((~same).sum())
                with np.errstate(invalid="ignore"):
                    inside = (~same) & (np.abs(x) < np.float32(2.0e9))
                bad_inside += int(inside.sum())
                smallest = min(smallest, float(np.abs(x[~same]).min()))
                if bad_inside:
                    print("%s: first difference at x = %r (bits %08x): gpu %r oracle %r" % (fn, x[i], int(x[i:i + 1].view(np.uint32)[0]), g[i], w[i]), flush=True)
        rec[fn] = {"inputs": 1 << 32, "differences": bad, "differences_below_2e9": bad_inside, "smallest_differing_magnitude": None if bad == 0 else smallest,
                   "seconds": round(time.perf_counter() - t0, 1)}
        print(fn, rec[fn], flush=True)
    rng = np.random.default_rng(4)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, 0.5, 2.0, 1e-45, -1e-45, 1.1754944e-38, 3.4028235e38, -3.4028235e38, 1e-20, 1e20, 0.99999994, 1.0000001,
                        3.1415927, 1.5707964, 2.2, 0.45454547], np.float32)
    vals = np.concatenate([special, rng.uniform(-4, 4, 2000), np.exp(rng.uniform(-80, 80, 1037)) * rng.choice([-1, 1], 1037), rng.integers(0, 1 << 32, 1037, dtype=np.uint64).astype(np.uint32).view(np.float32)]).astype(np.float32)
    assert vals.size == 4096
    a, b = np.meshgrid(vals, vals, indexing="ij")
    a, b = np.ascontiguousarray(a.ravel()), np.ascontiguousarray(b.ravel())
    for fn in ("atan2", "pow", "div"):
        g = gpu.debug_eval(fn, np.stack([a, b], 1), 1)[:, 0]
        w = oracle_eval(fn, a, b) if fn != "div" else a / b
        same = (g.view(np.uint32) == w.view(np.uint32)) | (np.isnan(g) & np.isnan(w))
        rec[fn] = {"inputs": int(a.size), "differences": int((~same).sum())}
        if not same.all():
            i = int(np.argmin(same))
            print("%s: first difference at (%r, %r): gpu %r oracle %r" % (fn, a[i], b[i], g[i], w[i]), flush=True)
        print(fn, rec[fn], flush=True)
    out_path = os.path.join(os.path.dirname(__file__), "..", "gpurun_out", "libm_exhaustive.json")
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    with open(out_path, "w") as f:
        json.dump(rec, f, indent=1)
    # sin / cos / tan reduce their argument through an int conversion: beyond |x| = 2^31 * pi / 2 = 3.37e9 the conversion saturates differently on
    # the two machines (the renderer's arguments are 2 pi u and a field of view: the contract's domain, pt_device.h); everything else: all inputs
    assert all(rec[k]["differences"] == 0 for k in ("atan", "asin", "log", "exp", "sqrt", "atan2", "pow", "div")), rec
    assert all(rec[k]["differences_below_2e9"] == 0 for k in ("sin", "cos", "tan")), rec


def test_rng_kat_on_device(gpu, orc):
    seeds = np.array([[0, 0], [1919, 1079], [1, 0], [0, 1], [255, 255], [7, 11], [4095, 4095]], np.uint32)
    out = gpu.debug_eval("rng", seeds.view(np.float32), 5)
    want_state0 = [1576399551, 2688469361, 3231205618, 1569133783, 1606964575]  # SURVEY 8(a6), reference random.hpp
    assert out[:5, 0].view(np.uint32).tolist() == want_state0
    assert out[0, 4:5].view(np.uint32)[0] == 595458768 and out[1, 4:5].view(np.uint32)[0] == 1139548942
    for i, (u, v) in enumerate(seeds):
        s = orc.rng_init(int(u), int(v))
        assert out[i, 0:1].view(np.uint32)[0] == s
        for k in range(3):
            f, s = orc.rng_next(s)
            assert out[i, 1 + k] == np.float32(f)


def test_frame_math_bitwise(gpu, orc):
    rng = np.random.default_rng(5)
    n = rng.normal(size=(2000, 3))
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    n[0] = [0, 1, 0]
    n[1] = np.float32(1 / np.sqrt(3))
    w = rng.normal(size=(2000, 3))
    inp = np.concatenate([n, w], 1).astype(np.float32)
    out = gpu.debug_eval("frame", inp, 12)
    for i in range(0, 2000, 7):
        t, b = orc.onb(inp[i, :3])
        l = orc.to_local(t, b, inp[i, :3], inp[i, 3:])
        g = orc.to_world(t, b, inp[i, :3], l)
        assert_bitwise(out[i], np.concatenate([t, b, l, g]), "frame %d" % i)


def test_sample_disney_bitwise(gpu, orc, scene_io):
    import json

    rng = np.random.default_rng(1234)
    mats = [scene_io.material(roughness=1.0), scene_io.material(metallic=1.0, roughness=0.2),
            scene_io.material(specular_transmission=1.0, roughness=0.0, specular_transmission_roughness=0.01),
            scene_io.material(metallic=1.0, roughness=0.1, anisotropic=1.0, specular=0.0),
            scene_io.material(clearcoat=1.0, clearcoat_gloss=0.9, base_color=[0.0272, 0.112622, 0.8]),
            scene_io.material(clearcoat=1.0, clearcoat_gloss=0.1, sheen=0.8, sheen_tint=0.5, base_color=[0.7, 0.3, 0.2]),
            scene_io.material(metallic=0.3, clearcoat=0.6, specular_transmission=0.5, specular_transmission_roughness=0.3, roughness=0.4, sheen=0.2),
            scene_io.material(base_color=[0, 0, 0], roughness=0.75), scene_io.MAT_DEFAULT.copy()]
    for name in ("car", "cornell-box", "cube", "dragon", "mitsuba"):
        _, ms = scene_io.parse_scene(os.path.join(os.path.dirname(B.HEADER_PATH), "..", "assets", name + ".json"))
        mats += [m for _, m, _ in ms]
    rows = []
    for m in mats:
        for k in range(120):
            wo = rng.normal(size=3)
            wo /= np.linalg.norm(wo)
            if k % 3 == 0:
                wo[2] = abs(wo[2])
            if k % 17 == 0:
                wo = np.array([1.0, 0.0, 1e-4 * (k % 5)])  # grazing
                wo /= np.linalg.norm(wo)
            st = np.uint32(rng.integers(0, 2 ** 32))
            lobe = np.int32([-1, 3, 0, 2][k % 4])
            rows.append(np.concatenate([m, wo.astype(np.float32), np.array([st]).view(np.float32), np.array([lobe]).view(np.float32)]))
    inp = np.stack(rows).astype(np.float32)
    out = gpu.debug_eval("sample_disney", inp, 9)
    lobes = set()
    for i in range(inp.shape[0]):
        r = orc.sample_disney(inp[i, :17], inp[i, 17:20], int(inp[i, 20:21].view(np.uint32)[0]), int(inp[i, 21:22].view(np.int32)[0]))
        want = np.concatenate([r["f"], r["wi"], [r["pdf"]]]).astype(np.float32)
        assert_bitwise(out[i, :7], want, "sample_disney row %d" % i)
        assert int(out[i, 7:8].view(np.int32)[0]) == r["lobe"] and int(out[i, 8:9].view(np.uint32)[0]) == r["state"]
        lobes.add(r["lobe"])
    assert lobes == {0, 1, 2, 3}


def test_closest_hit_bitwise(gpu, orc, cornell):
    _upload(gpu, cornell)
    S = orc.Scene(cornell["flat"])
    P = cornell["flat"]["positions"].reshape(-1, 3, 3)
    rng = np.random.default_rng(77)
    n = 20000
    o = rng.uniform(-1.5, 1.5, (n, 3)) + [0, 1, 0]
    k = rng.integers(len(P), size=n // 2)
    bary = rng.dirichlet([1, 1, 1], n // 2)
    o[: n // 2] = (bary[:, :, None] * P[k]).sum(1)  # secondary-like origins on surfaces
    d = rng.normal(size=(n, 3))
    d[::11, 1] = 0
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    inp = np.concatenate([o, d], 1).astype(np.float32)
    out = gpu.debug_eval("closest_hit", inp, 5)
    hits = 0
    for i in range(0, n, 5):
        ok, t, u, v, prim = S.intersect(inp[i, :3], inp[i, 3:], use_bvh=(i % 2 == 0))
        assert bool(out[i, 0]) == ok
        if ok:
            assert_bitwise(out[i, 1:4], np.float32([t, u, v]), "hit %d" % i)
            assert int(out[i, 4:5].view(np.int32)[0]) == prim
            hits += 1
    assert hits > 1500


# ---------------------------------------------------------------------------------------------------------------------
# images
# ---------------------------------------------------------------------------------------------------------------------

def test_cube_image_bitwise_c1(gpu, orc, cube, scene_io):
    """BASELINE config C1 (cube.json, 256x256, 16 spp, depth 4) at full size: textured, sky environment."""
    tex = scene_io.checker_texture()
    env = B.make_env(use_auto=True, intensity=1.0)
    _upload(gpu, cube, textures=[tex], mesh_textures=[0], env=env)
    W = H = 256
    cam = _cam(cube, W, H)
    gpu.set_option("count", 1)
    rgb, rgba = gpu.render(cam, W, H, 16, 4, want_rgba8=True)
    st = gpu.stats()
    gpu.set_option("count", 0)
    S = orc.Scene(cube["flat"])
    want, want8, cnt = S.render(_ocam(orc, cam), orc.make_env(use_auto=True, intensity=1.0), W, H, 16, 4, want_rgba8=True, want_counters=True)
    assert_bitwise(rgb, want, "cube C1 image")
    np.testing.assert_array_equal(rgba, want8)
    for k in ("samples", "rays", "scatters", "env_misses", "nan_retries"):
        assert st[k] == cnt[k], k
    assert rgb.std() > 0.01
    # short jobs (every slot renders one pixel and dies: all wind-down) through both kernel instances, plain and instrumented: a
    # register-spilling build of the instrumented instance once rendered hundreds of wrong pixels exactly here
    for spp, depth in ((1, 2), (1, 3), (2, 2)):
        want, _, _ = S.render(_ocam(orc, cam), orc.make_env(use_auto=True, intensity=1.0), W, H, spp, depth)
        for count in (0, 1):
            gpu.set_option("count", count)
            got, _ = gpu.render(cam, W, H, spp, depth)
            gpu.set_option("count", 0)
            assert_bitwise(got, want, "cube %d spp depth %d count=%d" % (spp, depth, count))


def test_cornell_image_bitwise_and_golden(gpu, orc, cornell):
    from conftest import GOLDEN

    env = B.make_env(color=(1, 1, 1), intensity=0.0)
    _upload(gpu, cornell, env=env)
    W = H = 48
    cam = _cam(cornell, W, H)
    rgb, _ = gpu.render(cam, W, H, 16, 16)
    assert_bitwise(rgb, np.load(os.path.join(GOLDEN, "cornell_48x48_16spp_d16_oracle.npy")), "cornell golden")
    W = H = 128
    cam = _cam(cornell, W, H)
    gpu.set_option("count", 1)
    rgb, rgba = gpu.render(cam, W, H, 32, 16, want_rgba8=True)
    st = gpu.stats()
    gpu.set_option("count", 0)
    S = orc.Scene(cornell["flat"])
    want, want8, cnt = S.render(_ocam(orc, cam), orc.make_env(color=(1, 1, 1), intensity=0.0), W, H, 32, 16, want_rgba8=True, want_counters=True)
    assert_bitwise(rgb, want, "cornell 128x128x32")
    np.testing.assert_array_equal(rgba, want8)
    for k in ("samples", "rays", "scatters", "nan_retries"):
        assert st[k] == cnt[k], k
    assert st["kernel_ms"] > 0 and st["launches"] == 2  # cost pre-pass + main launch over the cost-ordered pixel queue


def _queue_classes(ids, cost, q, W, H, R=2):
    """cost_bucket(pixel_cost()) of pt_kernel.hip for the queue entries q: the pre-pass class of a pixel (16 per doubling of the time its
    pre-pass samples took), replaced by the mean over the neighbours within +-10 classes if that is larger, in buckets of 4 from 230 down."""
    img = np.zeros((H, W), np.int32)
    img.reshape(-1)[ids] = cost
    pad = np.pad(img, R)
    sh = np.stack([pad[dy:dy + H, dx:dx + W] for dy in range(2 * R + 1) for dx in range(2 * R + 1)])
    ok = (sh != 0) & (np.abs(sh - img[None]) <= 10)
    cnt = ok.sum(0)
    mean = np.where(cnt > 0, (np.where(ok, sh, 0).sum(0) + cnt - 1) // np.maximum(cnt, 1), img)
    est = np.maximum(img, mean).reshape(-1)
    return np.clip((230 - est[q]) // 4, 0, 31)


def test_cost_ordered_queue_properties(gpu, cornell):
    """The cost pre-pass + counting sort (default schedule from 32 spp): the queue is a permutation of the shard's pixels in
    non-increasing order of their de-noised time class (most expensive first), stable within a class."""
    env = B.make_env(color=(1, 1, 1), intensity=0.0)
    _upload(gpu, cornell, env=env)
    W, H = 96, 80
    cam = _cam(cornell, W, H)
    gpu.set_pixel_shard(1, 2, 16)
    gpu.set_option("whole", 0)
    try:
        gpu.render(cam, W, H, 40, 16)
    finally:
        gpu.set_option("whole", -1)
    assert gpu.stats()["launches"] == 2 and gpu.stats()["prepass_ms"] > 0
    q, ids, cost = gpu.read_queue(W * H)
    gpu.set_pixel_shard(0, 1, 16)
    own = B.shard_pixels(W, H, 16, 1, 2)
    np.testing.assert_array_equal(ids, own)
    np.testing.assert_array_equal(np.sort(q), np.sort(own))
    assert cost.min() >= 1  # a class per pixel of the shard (0 = not this rank's); the classes are wall-clock times: nothing else about their values is asserted
    cls = _queue_classes(ids, cost, q, W, H)
    assert np.all(np.diff(cls) >= 0), "queue buckets must be non-decreasing (bucket 0 = most expensive first)"
    # stable within a class: input order is kept
    pos = np.empty(W * H, np.int64)
    pos[ids] = np.arange(ids.size)
    for c in np.unique(cls):
        assert np.all(np.diff(pos[q[cls == c]]) > 0)


def test_tier_schedule_bitwise(gpu, orc, cornell):
    """Whole-pixel schedule by cost class (pt_kernel.hip, TIERS): a launch whose pixels all have a path slot may hand out pixels instead
    of (pixel, chunk) tickets, every wave serving one cost class.  Forced on, forced off and left to the plan: the same image bit
    for bit (== the oracle's), and the tier table partitions the queue among workgroups that exist."""
    env = dict(color=(1, 1, 1), intensity=0.0)
    _upload(gpu, cornell, env=B.make_env(**env))
    W, H, spp = 64, 48, 40
    cam = _cam(cornell, W, H)
    want, _, cnt = orc.Scene(cornell["flat"]).render(_ocam(orc, cam), orc.make_env(**env), W, H, spp, 16, want_counters=True)
    try:
        for whole, count in ((1, 0), (1, 1), (0, 0), (-1, 0)):
            gpu.set_option("whole", whole)
            gpu.set_option("count", count)
            got, _ = gpu.render(cam, W, H, spp, 16)
            st = gpu.stats()
            gpu.set_option("count", 0)
            assert_bitwise(got, want, "whole=%d count=%d" % (whole, count))
            if count:
                for k in ("samples", "rays", "scatters"):
                    assert st[k] == cnt[k], (whole, k)
            tiers = gpu.read_tiers()
            if whole == 1:
                assert st["whole_pixels"] == W * H and tiers, (st["whole_pixels"], tiers)
                q0 = 0
                for t in tiers:  # consecutive queue ranges, workgroup ranges one after the other, pixels per wave within the menu
                    assert t["q0"] == q0 and t["pixels"] > 0 and 4 <= t["per_wave"] <= 104
                    assert t["waves"] * t["per_wave"] >= t["pixels"] > (t["waves"] - 1) * t["per_wave"]
                    q0 += t["pixels"]
                assert q0 == W * H
                assert [t["wave0"] for t in tiers] == list(np.cumsum([0] + [t["waves"] for t in tiers[:-1]]))
                assert tiers[-1]["wave0"] + tiers[-1]["waves"] <= st["grid"]
                assert [t["cost_class"] for t in tiers] == sorted(t["cost_class"] for t in tiers)
            if whole == 0:
                assert st["whole_pixels"] == 0 and not tiers
        # tiny frames: fewer pixels than a wave has lanes, one pixel, a ragged size (always planned: <= 16 pixels per resident wave)
        gpu.set_option("whole", -1)
        S = orc.Scene(cornell["flat"])
        for w, h in ((1, 1), (3, 2), (17, 5)):
            cam_s = _cam(cornell, w, h)
            got, _ = gpu.render(cam_s, w, h, 64, 16)
            assert gpu.stats()["launches"] == 2 and gpu.read_tiers(), (w, h)
            want_s, _, _ = S.render(_ocam(orc, cam_s), orc.make_env(**env), w, h, 64, 16)
            assert_bitwise(got, want_s, "tiers, %dx%d" % (w, h))
        # a frame that fills a good part of the chip, planned (forced) against the ring schedule: every pixel, every bit
        # (workgroups beyond the plan once took tickets from the first tier's counter: its pixels stayed black now and then)
        W2, H2 = 320, 200
        cam2 = _cam(cornell, W2, H2)
        imgs = {}
        for whole in (1, 0):
            gpu.set_option("whole", whole)
            imgs[whole], _ = gpu.render(cam2, W2, H2, 48, 16)
            assert (gpu.stats()["whole_pixels"] != 0) == (whole == 1)
        assert_bitwise(imgs[1], imgs[0], "tiers vs ring, %dx%d" % (W2, H2))
        # more pixels than path slots, plan forced: the slots of a tier take one pixel after the other (rounds)
        W3, H3 = 1280, 720
        cam3 = _cam(cornell, W3, H3)
        for whole in (1, 0):
            gpu.set_option("whole", whole)
            imgs[whole], _ = gpu.render(cam3, W3, H3, 32, 16)
            if whole == 1:
                t3 = gpu.read_tiers()
                assert t3 and sum(t["waves"] * t["per_wave"] for t in t3) < W3 * H3
        assert_bitwise(imgs[1], imgs[0], "tiers with rounds vs ring, %dx%d" % (W3, H3))
    finally:
        gpu.set_option("whole", -1)
        gpu.set_option("count", 0)


def test_material_coverage_image_bitwise(gpu, orc, scene_io, procedural):
    """All four lobes + sheen + textures + emitter in one small scene (the car.json material set on spheres)."""
    _, car = scene_io.parse_scene(os.path.join(os.path.dirname(B.HEADER_PATH), "..", "assets", "car.json"))
    mats = [(n, (m if n != "Ground" else m), "") for n, m, _ in car]
    meshes = []
    for i, (name, m, _) in enumerate(mats):
        if name == "Ground":
            meshes.append((name, procedural.quad((-6, 0, -6), (-6, 0, 6), (6, 0, 6), (6, 0, -6), (0, 1, 0), uv=True)))
        elif name == "Light":
            meshes.append((name, procedural.quad((-2, 4, -2), (2, 4, -2), (2, 4, 2), (-2, 4, 2), (0, -1, 0))))
        else:
            a = 2 * np.pi * i / len(mats)
            meshes.append((name, procedural.uv_sphere((2.2 * np.cos(a), 0.5, 2.2 * np.sin(a)), 0.5, nu=24, nv=12)))
    ents = scene_io.build_entities(meshes, mats)
    gi = [n for n, _, _ in mats].index("Ground")
    tex = scene_io.checker_texture(32, 32, 4)
    flat = scene_io.flatten_scene(ents, mats, {gi: tex})
    env = dict(use_auto=True, intensity=0.6)
    gpu.upload_scene(ents, [m for _, m, _ in mats], textures=[tex], mesh_textures=[0 if mid == gi else -1 for _, mid in ents], env=B.make_env(**env))
    W, H = 96, 64
    cam = mkcam([0, 3.5, 6.5], [0, 0.4, 0], [0, 1, 0], 45, W, H)
    rgb, rgba = gpu.render(cam, W, H, 24, 16, want_rgba8=True)
    S = orc.Scene(flat)
    want, want8, cnt = S.render(_ocam(orc, cam), orc.make_env(**env), W, H, 24, 16, want_rgba8=True, want_counters=True)
    assert_bitwise(rgb, want, "material coverage")
    np.testing.assert_array_equal(rgba, want8)
    # environment map path (device.cu:23-39,138-139): same scene under a synthetic 8-bit lat-long map
    yy, xx = np.mgrid[0:16, 0:32]
    envmap = ((xx * 8) | ((yy * 16) << 8) | (((xx + yy) * 5) << 16) | (0xFF << 24)).astype(np.uint32)
    gpu.set_environment(B.make_env(use_map=True, intensity=1.0, env_map=envmap))
    rgb, _ = gpu.render(cam, W, H, 8, 16)
    want, _, cnt = S.render(_ocam(orc, cam), orc.make_env(use_map=True, intensity=1.0, env_map=envmap), W, H, 8, 16, want_counters=True)
    assert cnt["env_misses"] > 0
    assert_bitwise(rgb, want, "environment map")


def test_chunked_sharded_and_material_sweep(gpu, orc, cornell):
    env = B.make_env(color=(1, 1, 1), intensity=0.0)
    _upload(gpu, cornell, env=env)
    W, H = 80, 56
    cam = _cam(cornell, W, H)
    full, _ = gpu.render(cam, W, H, 23, 16)
    # resumable spp chunks carry (rng, accum) per pixel: identical image (RNG stream is sequential per pixel)
    gpu.set_option("spp_per_launch", 5)
    chunked, _ = gpu.render(cam, W, H, 23, 16)
    assert gpu.stats()["launches"] == 1  # the wavefront kernel walks its (pixel, chunk) tickets inside one persistent launch
    gpu.set_option("spp_per_launch", 0)
    assert_bitwise(chunked, full, "chunked == single launch")
    # the three schedules of the wavefront kernel give one image: cost-ordered queue (pre-pass + sort + main launch, default from
    # 32 spp), FIFO ring of 64-spp tickets with a shrinking tail, ring with a fixed chunk
    by_cost, _ = gpu.render(cam, W, H, 70, 16)
    assert gpu.stats()["launches"] == 2
    gpu.set_option("schedule", 0)
    gpu.set_option("chunk_spp", 16)
    ring, _ = gpu.render(cam, W, H, 70, 16)
    assert gpu.stats()["launches"] == 1
    gpu.set_option("chunk_tail_min", 0)
    ring_fixed, _ = gpu.render(cam, W, H, 70, 16)
    gpu.set_option("chunk_tail_min", -1)
    gpu.set_option("chunk_spp", 64)
    gpu.set_option("schedule", 1)
    assert_bitwise(ring, by_cost, "FIFO ring == cost-ordered queue")
    assert_bitwise(ring_fixed, by_cost, "fixed chunks == cost-ordered queue")
    # schedule corner cases against the oracle: smallest spp that sorts (4 x prepass_spp), odd counts, a first chunk of 1 % / 100 %
    # of the samples, no halving tail, a 1-sample pre-pass, no neighbourhood smoothing, few slots per wave
    S = orc.Scene(cornell["flat"])
    oenv = orc.make_env(color=(1, 1, 1), intensity=0.0)
    # (whole = 0: these are options of the ring schedule; a frame this small would otherwise get a tier plan - the last two cases)
    for spp, opts in ((32, {"whole": 0}), (33, {"sticky_pct": 1, "whole": 0}), (41, {"sticky_pct": 100, "whole": 0}), (37, {"chunk_tail_min": 0, "sticky_pct": 50, "whole": 0}),
                      (36, {"prepass_spp": 1, "cost_radius": 0, "whole": 0}), (45, {"prepass_spp": 11, "chunk_tail_min": 3, "slots_per_wave": 64, "whole": 0}),
                      (32, {}), (71, {"prepass_spp": 3, "cost_radius": 0})):
        for k, v in opts.items():
            gpu.set_option(k, v)
        got, _ = gpu.render(cam, W, H, spp, 16)
        assert gpu.stats()["launches"] == 2, (spp, opts)
        assert (gpu.stats()["whole_pixels"] != 0) == ("whole" not in opts), (spp, opts)
        for k, v in {"sticky_pct": -1, "chunk_tail_min": -1, "prepass_spp": 0, "cost_radius": 2, "slots_per_wave": 0, "whole": -1}.items():
            gpu.set_option(k, v)
        want, _, _ = S.render(_ocam(orc, cam), oenv, W, H, spp, 16)
        assert_bitwise(got, want, "schedule %r at %d spp" % (opts, spp))
    # topped-up shading passes (option "tune0": a pass with idle lanes also takes entries of the other queue from this many on)
    for k in (1, 8, 24, 64):
        gpu.set_option("tune0", k)
        got, _ = gpu.render(cam, W, H, 70, 16)
        gpu.set_option("tune0", 0)
        assert_bitwise(got, by_cost, "topped-up shading passes, threshold %d" % k)
    # pixel-tile shards are disjoint: the sum over ranks (what the RCCL reduce computes) is the 1-GPU image bit-for-bit
    acc = np.zeros_like(full)
    for r in range(3):
        gpu.set_pixel_shard(r, 3, 16)
        part, _ = gpu.render(cam, W, H, 23, 16)
        if r == 1:  # the same shard with the cost-ordered queue and its tier plan: identical
            part70, _ = gpu.render(cam, W, H, 70, 16)
            assert gpu.stats()["launches"] == 2 and gpu.stats()["whole_pixels"] != 0
        own = np.zeros(W * H, bool)
        own[B.shard_pixels(W, H, 16, r, 3)] = True
        own = own.reshape(H, W)[::-1]  # framebuffer rows are flipped (device.cu:251)
        assert not part[~own].any()
        if r == 1:
            assert not part70[~own].any()
            assert_bitwise(part70[own], by_cost[own], "shard 1/3 with a tier plan == the full frame on its tiles")
        acc += part
    gpu.set_pixel_shard(0, 1, 16)
    assert_bitwise(acc, full, "sum of shards == full")
    # the same through the cost-ordered schedule (each rank sorts its own shard; the cost image is zero where it owns nothing)
    full36, _ = gpu.render(cam, W, H, 36, 16)
    acc = np.zeros_like(full36)
    for r in range(2):
        gpu.set_pixel_shard(r, 2, 16)
        part, _ = gpu.render(cam, W, H, 36, 16)
        assert gpu.stats()["launches"] == 2
        acc += part
    gpu.set_pixel_shard(0, 1, 16)
    assert_bitwise(acc, full36, "sum of cost-ordered shards == full")
    # material hot-swap without BVH rebuild (reset_field, application.cpp:297-304)
    mats = np.stack(_mats(cornell)).copy()
    mats[1, 7] = 0.9  # sphere roughness
    mats[0, 0:3] = [0.9, 0.2, 0.1]
    gpu.set_materials(mats)
    swept, _ = gpu.render(cam, W, H, 8, 16)
    S = orc.Scene(cornell["flat"])
    S.set_materials(mats)
    want, _, _ = S.render(_ocam(orc, cam), orc.make_env(color=(1, 1, 1), intensity=0.0), W, H, 8, 16)
    assert_bitwise(swept, want, "after pt_set_materials")


def test_empty_rank_and_world8_sum(gpu, cornell):
    """A rank that owns no tile (64x64 / tile 16 has 16 tiles on 7 diagonals: rank 7 of 8 gets none) renders a zero frame without
    launching a kernel - at 40 spp the cost-ordered schedule with a tier plan would otherwise read a table nobody wrote (round-3
    advisor finding) - and the eight shards still sum to the full frame."""
    _upload(gpu, cornell, env=B.make_env(color=(1, 1, 1), intensity=0.0))
    W = H = 64
    cam = _cam(cornell, W, H)
    full, full8 = gpu.render(cam, W, H, 40, 16, want_rgba8=True)
    assert B.shard_pixels(W, H, 16, 7, 8).size == 0
    acc = np.zeros_like(full)
    acc8 = np.zeros_like(full8)
    try:
        for r in range(8):
            gpu.set_pixel_shard(r, 8, 16)
            part, part8 = gpu.render(cam, W, H, 40, 16, want_rgba8=True)
            st = gpu.stats()
            if r == 7:
                assert not part.any() and not part8.any() and st["launches"] == 0
                gpu.set_option("count", 1)  # the counted instance takes the same early exit
                part, _ = gpu.render(cam, W, H, 40, 16)
                gpu.set_option("count", 0)
                assert not part.any()
            else:
                assert st["launches"] == 2
            acc += part
            acc8 += part8  # disjoint pixels: 0 where not owned
    finally:
        gpu.set_pixel_shard(0, 1, 16)
    assert_bitwise(acc, full, "sum of 8 shards (one of them empty) == full")
    np.testing.assert_array_equal(acc8, full8)


def test_slab_test_forms_and_far_camera(gpu, orc, cornell):
    """The quad step computes slab distances as fma(plane, 1/d, -(o/d)) (round 4) and the host switches to (plane - o) * (1/d) for a camera
    far outside the scene (pt_api.cpp, box_exact): both forms must give the oracle's image - boxes are conservative either way - and so
    must a camera 3 000 units away (300 scene extents), where the fma form's error would exceed the boxes' padding."""
    _upload(gpu, cornell, env=B.make_env(color=(1, 1, 1), intensity=0.0))
    S = orc.Scene(cornell["flat"])
    oenv = orc.make_env(color=(1, 1, 1), intensity=0.0)
    W, H = 96, 72
    cam = _cam(cornell, W, H)
    want, _, _ = S.render(_ocam(orc, cam), oenv, W, H, 40, 16)
    try:
        for form in (0, 1, -1):
            gpu.set_option("box_exact", form)
            got, _ = gpu.render(cam, W, H, 40, 16)
            assert_bitwise(got, want, "slab form %d" % form)
            gpu.set_option("groups", 2)  # the group walk has both forms too
            got, _ = gpu.render(cam, W, H, 40, 16)
            gpu.set_option("groups", 1)
            assert_bitwise(got, want, "slab form %d, group walk" % form)
    finally:
        gpu.set_option("box_exact", -1)
        gpu.set_option("groups", 1)
    far = mkcam([3000.0, 1.0, 0.0], [0.0, 1.0, 0.0], [0.0, 1.0, 0.0], 0.05, W, H)
    got, _ = gpu.render(far, W, H, 40, 16)
    want, _, _ = S.render(_ocam(orc, far), oenv, W, H, 40, 16)
    assert got.max() > 0.0
    assert_bitwise(got, want, "camera 3000 units away (automatic: subtracting form)")


def test_axis_parallel_cameras(gpu, orc, cornell):
    """Cameras that look exactly along +-x, +-y, +-z from points whose coordinates are not 0: in the middle row AND the middle column the
    jitter is absorbed when the direction is formed, so those camera rays have one (at the centre: two) direction components of exactly 0 -
    1/d = inf.  Both slab forms, the group walk and the lane-per-pixel kernel must give the oracle's image (the fma form culled boxes that
    straddle 0 on such an axis before ray_inv clamped the reciprocal: csrc/pt_trace.h)."""
    _upload(gpu, cornell, env=B.make_env(color=(1, 1, 1), intensity=0.35))
    S = orc.Scene(cornell["flat"])
    oenv = orc.make_env(color=(1, 1, 1), intensity=0.35)
    V = np.asarray(cornell["flat"]["positions"], np.float32).reshape(-1, 3)
    lo, hi = V.min(0), V.max(0)
    centre = ((lo + hi) * 0.5 + np.array([0.25, 0.125, 0.5], np.float32) * (hi - lo) * 0.25).astype(np.float32)
    # a narrow field of view makes the window wide: |jitter| / H * 2 tan(fov / 2) < ulp(origin) / 2 holds for ~3e-4 of the samples of the two
    # middle rows / columns at 0.5 degrees, i.e. a handful of such rays per camera at 32 x 32 x 512 (with infinity for the clamp,
    # `make variant NAME=noclamp FLAGS='-DPT_INV_MAX=__builtin_inff\(\)'`, every camera of this test fails)
    W = H = 32
    n_zero = 0
    try:
        for axis in range(3):
            for sign in (-1.0, 1.0):
                frm = centre.copy()
                frm[axis] += sign * 0.45 * (hi - lo)[axis]  # inside the scene's box, looking through its centre
                up = [0.0, 0.0, 0.0]
                up[(axis + 1) % 3] = 1.0
                cam = mkcam([float(x) for x in frm], [float(x) for x in centre], up, 0.5, W, H)
                want, _, _ = S.render(_ocam(orc, cam), oenv, W, H, 512, 8)
                for opts in ((), (("box_exact", 1),), (("groups", 2),), (("kernel", 1),)):
                    for k, v in opts:
                        gpu.set_option(k, v)
                    got, _ = gpu.render(cam, W, H, 512, 8)
                    for k, _ in opts:
                        gpu.set_option(k, {"box_exact": -1, "groups": 1, "kernel": 2}[k])
                    assert_bitwise(got, want, "camera along %s%s, options %s" % ("-+"[sign > 0], "xyz"[axis], opts))
                n_zero += 1
    finally:
        for k, v in (("box_exact", -1), ("groups", 1), ("kernel", 2)):
            gpu.set_option(k, v)
    assert n_zero == 6


def test_nan_retry_path(gpu, orc, cornell):
    """device.cu:196-201: a BSDF value that is NaN or infinite sends the path back to the same hit with fresh draws.  No config and no
    finite material reaches that branch (nan_retries == 0 in every census), so it is driven here with base colours of NaN, infinity and
    3e38 (which overflows on the way: infinite throughput, NaN pixels): millions of retries per frame, the 64-retry safety net, NaN
    propagation into the frame - image, NaN pattern and work counters must be the oracle's, in every kernel."""
    _upload(gpu, cornell, env=B.make_env(color=(1, 1, 1), intensity=0.5))
    S = orc.Scene(cornell["flat"])
    oenv = orc.make_env(color=(1, 1, 1), intensity=0.5)
    base = np.stack(_mats(cornell)).astype(np.float32)
    W = H = 64
    cam = _cam(cornell, W, H)
    total = 0
    try:
        for who, val in ((0, np.nan), (1, np.inf), (3, 3e38), (0, 3e38), (3, np.nan)):
            mm = base.copy()
            mm[who, 0] = val
            mm[who, 1] = val
            gpu.set_materials(mm)
            S.set_materials(mm)
            want, _, cnt = S.render(_ocam(orc, cam), oenv, W, H, 32, 16, want_counters=True)
            total += cnt["nan_retries"]
            for opts in ((("count", 1),), (), (("groups", 2),), (("kernel", 1),)):
                for k, v in opts:
                    gpu.set_option(k, v)
                got, _ = gpu.render(cam, W, H, 32, 16)
                st = gpu.stats()
                for k, _ in opts:
                    gpu.set_option(k, {"count": 0, "groups": 1, "kernel": 2}[k])
                assert_bitwise(got, want, "base colour %r on material %d, options %s" % (val, who, opts))
                if opts and opts[0][0] == "count":
                    for k in ("samples", "rays", "scatters", "nan_retries"):
                        assert st[k] == cnt[k], (k, st[k], cnt[k])
    finally:
        gpu.set_materials(base)
        for k, v in (("count", 0), ("groups", 1), ("kernel", 2)):
            gpu.set_option(k, v)
    assert total > 1_000_000  # the branch was really taken


def test_lane_per_pixel_variant_bitwise(gpu, orc, cornell):
    """option kernel=1 (persistent lane-per-pixel scheduler) must produce the same bits as the default wavefront scheduler."""
    _upload(gpu, cornell, env=B.make_env(color=(1, 1, 1), intensity=0.0))
    W, H = 96, 72
    cam = _cam(cornell, W, H)
    a, _ = gpu.render(cam, W, H, 12, 16)
    gpu.set_option("kernel", 1)
    try:
        b, _ = gpu.render(cam, W, H, 12, 16)
        gpu.set_option("spp_per_launch", 5)
        c, _ = gpu.render(cam, W, H, 12, 16)
    finally:
        gpu.set_option("spp_per_launch", 0)
        gpu.set_option("kernel", 2)
    assert_bitwise(a, b, "kernel 1 == kernel 2")
    assert_bitwise(a, c, "kernel 1 chunked == kernel 2")
    S = orc.Scene(cornell["flat"])
    want, _, _ = S.render(_ocam(orc, cam), orc.make_env(color=(1, 1, 1), intensity=0.0), W, H, 12, 16)
    assert_bitwise(a, want, "vs oracle")


def test_error_paths(gpu):
    fresh = B.Context(0)
    cam = mkcam([0, 0, 3], [0, 0, 0], [0, 1, 0], 40, 8, 8)
    with pytest.raises(B.PtError, match="no geometries"):
        fresh.render(cam, 8, 8, 1, 1)
    # empty scene renders the environment only
    fresh.upload_scene([], np.zeros((0, 17), np.float32), env=B.make_env(color=(0.25, 0.5, 1.0), intensity=0.5))
    rgb, _ = fresh.render(cam, 8, 8, 3, 4)
    np.testing.assert_allclose(rgb, np.broadcast_to(np.float32([0.125, 0.25, 0.5]), (8, 8, 3)), rtol=1e-6)
    with pytest.raises(B.PtError):
        fresh.render(cam, 0, 8, 1, 1)
    fresh.close()


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE full sizes: size-independent properties + exact check of a random pixel subset
# ---------------------------------------------------------------------------------------------------------------------

def _subset_check(gpu_img, S, orc, cam, env, W, H, spp, depth, n_pix, seed):
    rng = np.random.default_rng(seed)
    ids = np.sort(rng.choice(W * H, n_pix, replace=False)).astype(np.uint32)
    sub = np.zeros((H, W, 3), np.float32)
    S.render(_ocam(orc, cam), env, W, H, spp, depth, pixel_list=ids, out=sub)
    ys, xs = (H - 1 - ids // W), ids % W
    assert_bitwise(gpu_img[ys, xs], sub[ys, xs], "pixel subset at full spp")


def _whole_frame_check(gpu, S, orc, cam, env, W, H, spp, depth, what):
    """EVERY pixel of the full-size frame at low spp, bit for bit: every pixel's camera rays and first bounces walk the deep BVH of
    the full-size scene (the full-spp comparison can only afford a pixel subset on the CPU)."""
    got, _ = gpu.render(cam, W, H, spp, depth)
    want, _, _ = S.render(_ocam(orc, cam), env, W, H, spp, depth)
    assert_bitwise(got, want, "%s: whole frame %dx%d at %d spp" % (what, W, H, spp))


def test_c2_cornell_full_size(gpu, orc, cornell):
    """BASELINE C2: cornell-box.json, 512x512, 256 spp, depth 16."""
    _upload(gpu, cornell, env=B.make_env(color=(1, 1, 1), intensity=0.0))
    W = H = 512
    cam = _cam(cornell, W, H)
    a, a8 = gpu.render(cam, W, H, 256, 16, want_rgba8=True)
    st = gpu.stats()
    b, _ = gpu.render(cam, W, H, 256, 16)
    assert_bitwise(a, b, "run-to-run determinism")
    gpu.set_option("spp_per_launch", 64)
    c, _ = gpu.render(cam, W, H, 256, 16)
    gpu.set_option("spp_per_launch", 0)
    assert_bitwise(a, c, "chunked == single launch at full size")
    assert np.isfinite(a).all() and a.min() >= 0
    S = orc.Scene(cornell["flat"])
    _subset_check(a, S, orc, cam, orc.make_env(color=(1, 1, 1), intensity=0.0), W, H, 256, 16, 1500, 3)
    # the two middle rows in full: the camera looks horizontally, so ~3e-5 of these rows' camera rays have a direction component of EXACTLY 0
    # (the jitter is absorbed when the direction is formed) - the case that the fma form of the slab test got wrong with an infinite
    # reciprocal (round 4: three pixels of this frame; ray_inv in csrc/pt_trace.h)
    ids = np.arange((H // 2 - 1) * W, (H // 2 + 1) * W, dtype=np.uint32)
    sub = np.zeros((H, W, 3), np.float32)
    S.render(_ocam(orc, cam), orc.make_env(color=(1, 1, 1), intensity=0.0), W, H, 256, 16, pixel_list=ids, out=sub)
    assert_bitwise(a[H // 2 - 1:H // 2 + 1], sub[H // 2 - 1:H // 2 + 1], "C2: the two middle rows at full spp (axis-parallel camera rays)")
    print("C2 kernel_ms=%.1f Msamples/s=%.1f vgprs=%d" % (st["kernel_ms"], W * H * 256 / st["kernel_ms"] / 1e3, st["vgprs"]))


def test_c4_dragon_standin_full_size(gpu, orc, scene_io, procedural):
    """BASELINE C4 on the documented stand-in: 1920x1080, 1024 spp, depth 16 (871 400-triangle 'dragon')."""
    _, mats = scene_io.parse_scene(os.path.join(os.path.dirname(B.HEADER_PATH), "..", "assets", "dragon.json"))
    meshes = procedural.dragon_standin()
    ents = scene_io.build_entities(meshes, mats)
    env = dict(color=(1, 1, 1), intensity=0.0)
    gpu.upload_scene(ents, [m for _, m, _ in mats], env=B.make_env(**env))
    W, H = 1920, 1080
    cam = mkcam([4, 2.5, 0], [0, 0.75, 0], [0, 1, 0], 50, W, H)
    a, _ = gpu.render(cam, W, H, 1024, 16)
    st = gpu.stats()
    assert np.isfinite(a).all()
    flat = scene_io.flatten_scene(ents, mats)
    S = orc.Scene(flat)
    _subset_check(a, S, orc, cam, orc.make_env(**env), W, H, 1024, 16, 600, 4)
    # the frame bench.py times is this frame: its checksum is the one bench.py asserts after its timed loop
    import json, zlib
    want_crc = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "c4_frame_crc.json")))["crc32_float3_frame"]
    assert zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF == want_crc, "C4 frame differs from tests/golden/c4_frame_crc.json (re-run bench.py --write-golden after an intended change)"
    _whole_frame_check(gpu, S, orc, cam, orc.make_env(**env), W, H, 4, 16, "C4")    # single launch, queue order
    _whole_frame_check(gpu, S, orc, cam, orc.make_env(**env), W, H, 32, 16, "C4")   # cost pre-pass + sorted queue + rings
    # hand-off stress (round-3 verdict item 5): every sample of every pixel a work item of its own - 64 chunks per pixel, 1.3e8
    # cross-wave hand-offs of (rng, accum) through the rings at 16 one-wave workgroups per CU (pt_kernel.hip, finish_chunk /
    # start_chunk) - against the frame whose pixels never change hands (one chunk).  A difference is a lost or stale publication.
    for k, v in (("schedule", 0), ("chunk_spp", 64)):
        gpu.set_option(k, v)
    try:
        whole64, _ = gpu.render(cam, W, H, 64, 16)
        assert gpu.stats()["launches"] == 1
        gpu.set_option("chunk_spp", 1)
        gpu.set_option("chunk_tail_min", 0)
        handed, _ = gpu.render(cam, W, H, 64, 16)
        ho_ms = gpu.stats()["kernel_ms"]
    finally:
        for k, v in (("schedule", 1), ("chunk_spp", 64), ("chunk_tail_min", -1)):
            gpu.set_option(k, v)
    assert_bitwise(handed, whole64, "C4 64 spp: 64 one-sample chunks per pixel == one chunk per pixel")
    print("C4 hand-off stress: %d hand-offs in %.0f ms" % (W * H * 63, ho_ms))
    # shards: rank 3 of 8 renders only its tiles, and exactly the full image's values there
    gpu.set_pixel_shard(3, 8, 16)
    part, _ = gpu.render(cam, W, H, 1024, 16)
    gpu.set_pixel_shard(0, 1, 16)
    own = np.zeros(W * H, bool)
    own[B.shard_pixels(W, H, 16, 3, 8)] = True
    own = own.reshape(H, W)[::-1]
    assert not part[~own].any()
    assert_bitwise(part[own], a[own], "shard 3/8 == full image on its tiles")
    print("C4 kernel_ms=%.1f Msamples/s=%.1f" % (st["kernel_ms"], W * H * 1024 / st["kernel_ms"] / 1e3))


def test_c3_mitsuba_standin_full_size(gpu, orc, scene_io, procedural):
    """BASELINE C3 on the documented stand-in: mitsuba.json, 1024x1024, 512 spp, depth 16, environment_auto (no emitter)."""
    _, mats = scene_io.parse_scene(os.path.join(os.path.dirname(B.HEADER_PATH), "..", "assets", "mitsuba.json"))
    ents = scene_io.build_entities(procedural.mitsuba_standin(), mats)
    env = dict(use_auto=True, intensity=1.0)
    gpu.upload_scene(ents, [m for _, m, _ in mats], env=B.make_env(**env))
    W = H = 1024
    cam = mkcam([4, 2.5, 0], [0, 0.75, 0], [0, 1, 0], 50, W, H)
    a, _ = gpu.render(cam, W, H, 512, 16)
    st = gpu.stats()
    S = orc.Scene(scene_io.flatten_scene(ents, mats))
    _subset_check(a, S, orc, cam, orc.make_env(**env), W, H, 512, 16, 800, 5)
    _whole_frame_check(gpu, S, orc, cam, orc.make_env(**env), W, H, 8, 16, "C3")
    assert a.mean() > 0.05
    print("C3 kernel_ms=%.1f Msamples/s=%.1f" % (st["kernel_ms"], W * H * 512 / st["kernel_ms"] / 1e3))


def test_c3_material_sweep_full_size(gpu, orc, scene_io, procedural):
    """The material sweep SURVEY 8(d) asks of C3 ("for BSDF coverage add a material sweep over metallic/clearcoat/transmission/sheen"; the
    reference's driver is test_loop / modify_sbt, application.hpp:89-108, application.cpp:329-360) at FULL size: 1024x1024, 512 spp, depth
    16 - the attribute is set on 'outside' and 'inside' through pt_set_materials (no BVH rebuild), a random pixel subset is compared
    with the oracle at full spp and, for the four-lobe variant, every pixel of the frame at 4 spp."""
    _, mats = scene_io.parse_scene(os.path.join(os.path.dirname(B.HEADER_PATH), "..", "assets", "mitsuba.json"))
    ents = scene_io.build_entities(procedural.mitsuba_standin(), mats)
    env = dict(use_auto=True, intensity=1.0)
    base = np.stack([m for _, m, _ in mats]).astype(np.float32)
    names = [n for n, _, _ in mats]
    gpu.upload_scene(ents, base, env=B.make_env(**env))
    W = H = 1024
    cam = mkcam([4, 2.5, 0], [0, 0.75, 0], [0, 1, 0], 50, W, H)
    S = orc.Scene(scene_io.flatten_scene(ents, mats))
    # material_data field indices (device_global.hpp:19-36): 4 metallic, 7 roughness, 9 sheen, 11 clearcoat, 14 transmission, 15 its roughness
    sweep = [("metallic 1, roughness .3", {4: 1.0, 7: 0.3}), ("clearcoat 1", {11: 1.0}), ("transmission .5, roughness .3", {14: 0.5, 15: 0.3}), ("sheen 1", {9: 1.0}),
             ("four lobes: metallic .3, clearcoat 1, transmission .5, sheen .5", {4: 0.3, 11: 1.0, 14: 0.5, 9: 0.5})]
    for k, (label, edits) in enumerate(sweep):
        mm = base.copy()
        for i, n in enumerate(names):
            if n != "ground":
                for f, v in edits.items():
                    mm[i, f] = v
        gpu.set_materials(mm)
        S.set_materials(mm)
        a, _ = gpu.render(cam, W, H, 512, 16)
        st = gpu.stats()
        assert np.isfinite(a).all(), label
        _subset_check(a, S, orc, cam, orc.make_env(**env), W, H, 512, 16, 300, 40 + k)
        if k == len(sweep) - 1:
            _whole_frame_check(gpu, S, orc, cam, orc.make_env(**env), W, H, 4, 16, "C3 " + label)
        print("C3 sweep %-62s kernel_ms=%.1f Msamples/s=%.1f" % (label, st["kernel_ms"], W * H * 512 / st["kernel_ms"] / 1e3))


def test_c5_car_standin_full_size(gpu, orc, scene_io, procedural):
    """BASELINE C5 on the documented stand-ins: car.json (12 materials: glass, clearcoat, anisotropic metal, textured ground,
    emitter), synthetic 2048x1024 HDR sky through stb's 8-bit tone map, 1920x1080, 4096 spp, depth 16, environment_use."""
    _, mats = scene_io.parse_scene(os.path.join(os.path.dirname(B.HEADER_PATH), "..", "assets", "car.json"))
    ents = scene_io.build_entities(procedural.car_standin(), mats)
    names = [n for n, _, _ in mats]
    gi = names.index("Ground")
    tex = scene_io.checker_texture(256, 256, 16)
    envmap = procedural.rgbe_to_ldr_rgba8(procedural.synthetic_sky_rgbe(2048, 1024))
    env = dict(use_map=True, intensity=1.0, env_map=envmap)
    gpu.upload_scene(ents, [m for _, m, _ in mats], textures=[tex], mesh_textures=[0 if mid == gi else -1 for _, mid in ents], env=B.make_env(**env))
    assert gpu.stats()["n_triangles"] > 1_500_000
    W, H = 1920, 1080
    cam = mkcam([0, 2, 5], [0, 0.5, 0], [0, 1, 0], 45, W, H)
    gpu.set_option("count", 1)
    a, _ = gpu.render(cam, W, H, 4096, 16)
    st = gpu.stats()
    gpu.set_option("count", 0)
    assert np.isfinite(a).all() and st["env_misses"] > 0 and st["samples"] == W * H * 4096
    S = orc.Scene(scene_io.flatten_scene(ents, mats, {gi: tex}))
    _subset_check(a, S, orc, cam, orc.make_env(**env), W, H, 4096, 16, 160, 6)
    _whole_frame_check(gpu, S, orc, cam, orc.make_env(**env), W, H, 4, 16, "C5")
    print("C5 (counted) kernel_ms=%.1f Msamples/s=%.1f rays/sample=%.2f" % (st["kernel_ms"], W * H * 4096 / st["kernel_ms"] / 1e3, st["rays"] / st["samples"]))


@pytest.mark.skipif(os.environ.get("PT_FULL_FRAME_PARITY") != "1", reason="opt-in (PT_FULL_FRAME_PARITY=1): ~8 minutes of oracle time on 16 threads")
def test_whole_frames_at_full_spp(gpu, orc, cornell, scene_io, procedural):
    """EVERY pixel of C2, C3 and C4 at the configs' full sample counts (C3 also under the five variants of the material sweep; C5: every 8th
    row at 4096 spp and every pixel for its first 256 spp) against the oracle, bit for bit - what the regular suite can only afford on
    pixel subsets (full spp) or at 4-32 spp (whole frames) - and C2 / C3 / C4 again through every other render path of the library.  Writes one record per config
    to gpurun_out/full_frame_parity.json (committed as profiles/rNN_full_frame_parity.json after a run)."""
    import json, time, zlib
    assets = os.path.join(os.path.dirname(B.HEADER_PATH), "..", "assets")
    out_path = os.path.join(os.path.dirname(__file__), "..", "gpurun_out", "full_frame_parity.json")
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    threads = _host_threads()
    records = []
    DEFAULTS = {"groups": 1, "whole": -1, "fallback": 0, "box_exact": -1, "kernel": 2, "schedule": 1, "count": 0, "quad": 1, "express_permille": -1, "chunk_spp": 64}
    variants = [("group walk for every traversal phase (groups=2)", [("groups", 2)]), ("no group walk (groups=0)", [("groups", 0)]),
                ("tier plan forced (whole=1)", [("whole", 1)]), ("no tier plan (whole=0)", [("whole", 0)]),
                ("50 per mille express pixels", [("whole", 0), ("express_permille", 50)]),
                ("140-VGPR fallback instance", [("fallback", 1)]), ("subtracting slab form (box_exact=1)", [("box_exact", 1)]),
                ("schedule 0, 16-sample chunks", [("schedule", 0), ("chunk_spp", 16)]),
                ("instrumented instance (count=1)", [("count", 1)]), ("instrumented instance, one-level walk (count=1, quad=0, groups=0)", [("count", 1), ("quad", 0), ("groups", 0)]),
                ("lane-per-pixel kernel", [("kernel", 1)])]

    def active_variants():
        return variants

    def check(name, S, cam, env_gpu, env_orc, W, H, spp, rows=None):
        got, _ = gpu.render(cam, W, H, spp, 16)
        ms = gpu.stats()["kernel_ms"]
        t0 = time.perf_counter()
        if rows is None:
            want, _, _ = S.render(_ocam(orc, cam), env_orc, W, H, spp, 16, threads=threads)
            a, b = got, want
        else:  # pixel ids count from the bottom row (B.shard_pixels convention); framebuffer row 0 is the top
            ids = (np.asarray(rows, np.uint32)[:, None] * np.uint32(W) + np.arange(W, dtype=np.uint32)[None, :]).ravel()
            sub = np.zeros((H, W, 3), np.float32)
            S.render(_ocam(orc, cam), env_orc, W, H, spp, 16, threads=threads, pixel_list=ids, out=sub)
            ys = H - 1 - np.asarray(rows)
            a, b = got[ys], sub[ys]
        dt = time.perf_counter() - t0
        same = bool((bits(a) == bits(b)).all())
        rec = {"config": name, "width": W, "height": H, "spp": spp, "pixels_compared": int(a.shape[0] * a.shape[1]), "bit_identical": same,
               "crc32_gpu": zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF, "crc32_oracle": zlib.crc32(np.ascontiguousarray(b).tobytes()) & 0xFFFFFFFF,
               "gpu_kernel_ms": round(ms, 2), "oracle_s": round(dt, 1), "oracle_threads": threads}
        records.append(rec)
        with open(out_path, "w") as f:
            json.dump(records, f, indent=1)
        print(rec, flush=True)
        assert same, "%s: whole frame at full spp differs from the oracle" % name
        # every other way this library can render the same frame (other walks, instances and schedules) - against the same oracle frame
        for label, opts in active_variants() if rows is None else ():
            for k, v in opts:
                gpu.set_option(k, v)
            try:
                alt, _ = gpu.render(cam, W, H, spp, 16)
                alt_ms = gpu.stats()["kernel_ms"]
            finally:
                for k, _ in opts:
                    gpu.set_option(k, DEFAULTS[k])
            ok = bool((bits(alt) == bits(want)).all())
            rec.setdefault("variants", []).append({"options": label, "bit_identical": ok, "gpu_kernel_ms": round(alt_ms, 2)})
            with open(out_path, "w") as f:
                json.dump(records, f, indent=1)
            print("   ", label, ok, "%.1f ms" % alt_ms, flush=True)
            assert ok, "%s with %s: whole frame at full spp differs from the oracle" % (name, label)

    # C2
    _upload(gpu, cornell, env=B.make_env(color=(1, 1, 1), intensity=0.0))
    check("C2 cornell-box 512x512x256", orc.Scene(cornell["flat"]), _cam(cornell, 512, 512), None, orc.make_env(color=(1, 1, 1), intensity=0.0), 512, 512, 256)
    # C3
    _, mats = scene_io.parse_scene(os.path.join(assets, "mitsuba.json"))
    ents = scene_io.build_entities(procedural.mitsuba_standin(), mats)
    env = dict(use_auto=True, intensity=1.0)
    gpu.upload_scene(ents, [m for _, m, _ in mats], env=B.make_env(**env))
    S3 = orc.Scene(scene_io.flatten_scene(ents, mats))
    cam3 = mkcam([4, 2.5, 0], [0, 0.75, 0], [0, 1, 0], 50, 1024, 1024)
    check("C3 mitsuba stand-in 1024x1024x512", S3, cam3, None, orc.make_env(**env), 1024, 1024, 512)
    # C3 under the material sweep (test_c3_material_sweep_full_size: subsets only): every pixel at full spp, all five variants
    base = np.stack([m for _, m, _ in mats]).astype(np.float32)
    all_variants, variants = variants, []
    for label, edits in (("metallic 1, roughness .3", {4: 1.0, 7: 0.3}), ("clearcoat 1", {11: 1.0}), ("transmission .5, roughness .3", {14: 0.5, 15: 0.3}), ("sheen 1", {9: 1.0}),
                         ("four lobes: metallic .3, clearcoat 1, transmission .5, sheen .5", {4: 0.3, 11: 1.0, 14: 0.5, 9: 0.5})):
        mm = base.copy()
        for i, (n, _, _) in enumerate(mats):
            if n != "ground":
                for f, v in edits.items():
                    mm[i, f] = v
        gpu.set_materials(mm)
        S3.set_materials(mm)
        check("C3 sweep: " + label, S3, cam3, None, orc.make_env(**env), 1024, 1024, 512)
    variants = all_variants
    # C4
    _, mats = scene_io.parse_scene(os.path.join(assets, "dragon.json"))
    ents = scene_io.build_entities(procedural.dragon_standin(), mats)
    env = dict(color=(1, 1, 1), intensity=0.0)
    gpu.upload_scene(ents, [m for _, m, _ in mats], env=B.make_env(**env))
    check("C4 dragon stand-in 1920x1080x1024", orc.Scene(scene_io.flatten_scene(ents, mats)), mkcam([4, 2.5, 0], [0, 0.75, 0], [0, 1, 0], 50, 1920, 1080), None,
          orc.make_env(**env), 1920, 1080, 1024)
    # C5: every 8th row
    _, mats = scene_io.parse_scene(os.path.join(assets, "car.json"))
    ents = scene_io.build_entities(procedural.car_standin(), mats)
    gi = [n for n, _, _ in mats].index("Ground")
    tex = scene_io.checker_texture(256, 256, 16)
    envmap = procedural.rgbe_to_ldr_rgba8(procedural.synthetic_sky_rgbe(2048, 1024))
    env = dict(use_map=True, intensity=1.0, env_map=envmap)
    gpu.upload_scene(ents, [m for _, m, _ in mats], textures=[tex], mesh_textures=[0 if mid == gi else -1 for _, mid in ents], env=B.make_env(**env))
    S5 = orc.Scene(scene_io.flatten_scene(ents, mats, {gi: tex}))
    cam5 = mkcam([0, 2, 5], [0, 0.5, 0], [0, 1, 0], 45, 1920, 1080)
    check("C5 car stand-in 1920x1080x4096, every 8th row", S5, cam5, None, orc.make_env(**env), 1920, 1080, 4096, rows=list(range(3, 1080, 8)))
    variants = []
    check("C5 car stand-in 1920x1080, every pixel, the first 256 of 4096 spp", S5, cam5, None, orc.make_env(**env), 1920, 1080, 256)


def test_one_quad_scene_leaf_root(gpu, orc, scene_io):
    """A scene of <= leaf_size triangles has a LEAF as BVH root (pt_bvh.cpp); the wavefront kernel used to spin on it (round-1 advisor
    finding).  One quad + sky, both kernels, against the oracle."""
    quad = dict(vertices=np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]], np.float32), normals=np.array([[0, 1, 0]] * 4, np.float32),
                texcoords=np.zeros((4, 2), np.float32), indices=np.array([[0, 1, 2], [0, 2, 3]], np.int32))
    mat = np.zeros((1, 17), np.float32)
    mat[0, :3] = (0.7, 0.6, 0.5); mat[0, 4] = 0.3; mat[0, 5] = 0.5; mat[0, 7] = 0.6; mat[0, 13] = 1.45
    ents = [(quad, 0)]
    env = B.make_env(use_auto=True, intensity=1.0)
    gpu.upload_scene(ents, mat, env=env)
    assert gpu.stats()["bvh_nodes"] == 0
    W, H = 96, 64
    cam = mkcam([0, 1.5, 3], [0, 0, 0], [0, 1, 0], 45, W, H)
    S = orc.Scene(scene_io.flatten_scene(ents, [("quad", mat[0], "")]))
    want, _, _ = S.render(_ocam(orc, cam), orc.make_env(use_auto=True, intensity=1.0), W, H, 16, 8)
    for k in (2, 1):
        gpu.set_option("kernel", k)
        got, _ = gpu.render(cam, W, H, 16, 8)
        assert_bitwise(got, want, "one quad, kernel %d" % k)
    gpu.set_option("kernel", 2)


def test_texture_lookup_edge_cases(gpu, orc, scene_io):
    """Texture coordinates outside [0, 1] (negative, > 1, exactly 0 and 1 at the corners), textures that are neither square nor powers of
    two, and a 1 x 1 texture: the nearest-texel lookup (device.cu:75-94 / tex2D with the reference's wrap mode) against the oracle on
    quads whose texcoords run from -1.25 to 2.5."""
    def quad(z, tc):
        return dict(vertices=np.array([[-1, 0, z - 1], [1, 0, z - 1], [1, 0, z + 1], [-1, 0, z + 1]], np.float32), normals=np.array([[0, 1, 0]] * 4, np.float32),
                    texcoords=np.asarray(tc, np.float32), indices=np.array([[0, 1, 2], [0, 2, 3]], np.int32))
    mat = np.zeros((3, 17), np.float32)
    mat[:, :3] = (0.7, 0.6, 0.5); mat[:, 5] = 0.5; mat[:, 7] = 0.6; mat[:, 13] = 1.45
    ents = [(quad(-2.2, [[-1.25, -0.5], [2.5, -0.5], [2.5, 1.75], [-1.25, 1.75]]), 0), (quad(0.0, [[0, 0], [1, 0], [1, 1], [0, 1]]), 1),
            (quad(2.2, [[0.3, 0.3], [0.4, 0.3], [0.4, 0.4], [0.3, 0.4]]), 2)]
    rng = np.random.default_rng(11)
    def tex(w, h):
        px = rng.integers(0, 256, (h, w, 3)).astype(np.uint32)
        return (px[..., 0] | (px[..., 1] << 8) | (px[..., 2] << 16) | (0xFF << 24)).astype(np.uint32)
    texs = [tex(5, 3), tex(7, 13), tex(1, 1)]
    gpu.upload_scene(ents, mat, textures=texs, mesh_textures=[0, 1, 2], env=B.make_env(use_auto=True, intensity=1.0))
    W, H = 160, 96
    cam = mkcam([0, 2.5, 5.5], [0, 0, 0], [0, 1, 0], 50, W, H)
    S = orc.Scene(scene_io.flatten_scene(ents, [("a", mat[0], ""), ("b", mat[1], ""), ("c", mat[2], "")], {0: texs[0], 1: texs[1], 2: texs[2]}))
    want, _, _ = S.render(_ocam(orc, cam), orc.make_env(use_auto=True, intensity=1.0), W, H, 24, 6)
    assert len(np.unique(want.reshape(-1, 3), axis=0)) > 1000
    for k in (2, 1):
        gpu.set_option("kernel", k)
        got, _ = gpu.render(cam, W, H, 24, 6)
        assert_bitwise(got, want, "textured quads, kernel %d" % k)
    gpu.set_option("kernel", 2)


def test_degenerate_and_coincident_geometry(gpu, orc, scene_io):
    """Coincident triangles (the same quad twice, with different materials: equal t - the lower global triangle id must win in every walk),
    zero-area triangles (two or three equal vertices: det = 0, never a hit), 200 copies of one small triangle (every centroid equal: the SAH
    builder's fallback split) and a needle a million times longer than wide - all three builders, both kernels, the group walk."""
    def mesh(v, idx):
        v = np.asarray(v, np.float32)
        return dict(vertices=v, normals=np.tile(np.array([[0, 1, 0]], np.float32), (len(v), 1)), texcoords=np.zeros((len(v), 2), np.float32), indices=np.asarray(idx, np.int32))
    q = [[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]]
    mat = np.zeros((4, 17), np.float32)
    mat[:, 5] = 0.5; mat[:, 7] = 0.6; mat[:, 13] = 1.45
    mat[0, :3] = (0.8, 0.1, 0.1); mat[1, :3] = (0.1, 0.8, 0.1); mat[2, :3] = (0.1, 0.1, 0.8); mat[3, :3] = (0.7, 0.7, 0.2)
    small = [[0.2, 0.3, 0.2], [0.4, 0.3, 0.2], [0.3, 0.3, 0.4]]
    ents = [(mesh(q, [[0, 1, 2], [0, 2, 3]]), 0), (mesh(q, [[0, 1, 2], [0, 2, 3]]), 1),                       # coincident quads
            (mesh([[0, 0.5, 0], [0, 0.5, 0], [1, 0.5, 0], [0.5, 0.5, 0.5]], [[0, 1, 2], [3, 3, 3], [0, 2, 2]]), 2),  # zero-area triangles
            (mesh(small * 200, np.arange(600).reshape(200, 3)), 3),                                             # 200 identical triangles
            (mesh([[-0.9, 0.2, -0.5], [0.9, 0.2000009, -0.5], [0.9, 0.2, -0.4999991]], [[0, 1, 2]]), 2)]          # a needle
    names = [("r", mat[0], ""), ("g", mat[1], ""), ("b", mat[2], ""), ("y", mat[3], "")]
    W, H = 128, 96
    cam = mkcam([0.4, 1.8, 2.6], [0, 0.1, 0], [0, 1, 0], 45, W, H)
    S = orc.Scene(scene_io.flatten_scene(ents, names))
    want, _, _ = S.render(_ocam(orc, cam), orc.make_env(use_auto=True, intensity=1.0), W, H, 24, 8)
    assert want[H // 2 + 10, W // 2, 0] > want[H // 2 + 10, W // 2, 1]  # the first (red) of the coincident quads wins
    try:
        for builder in (0, 1, 2):
            gpu.set_option("bvh_builder", builder)
            gpu.upload_scene(ents, mat, env=B.make_env(use_auto=True, intensity=1.0))
            for opts in ((), (("groups", 2),), (("kernel", 1),), (("count", 1), ("quad", 0), ("groups", 0))):
                for k, v in opts:
                    gpu.set_option(k, v)
                got, _ = gpu.render(cam, W, H, 24, 8)
                for k, _ in opts:
                    gpu.set_option(k, {"groups": 1, "kernel": 2, "count": 0, "quad": 1}[k])
                assert_bitwise(got, want, "degenerate scene, builder %d, options %s" % (builder, opts))
    finally:
        gpu.set_option("bvh_builder", 3)
        for k, v in (("groups", 1), ("kernel", 2), ("count", 0), ("quad", 1)):
            gpu.set_option(k, v)


def test_group_walk_bitwise(gpu, orc, cornell, scene_io, procedural):
    """Round 3: the group walk (eight lanes per ray over oct nodes, pt_kernel.hip traverse_groups) is what a SPARSE wave traverses
    with.  Closest hit does not depend on the visiting order, so forcing it on for every ray (groups = 2), leaving it to the sparse
    waves (1, default) and switching it off (0) must give one image, bit for bit - against the oracle on the small scenes, against
    each other on every pixel of the full-size C4 frame (deep tree: 871 k triangles)."""
    env = dict(color=(1, 1, 1), intensity=0.0)
    _upload(gpu, cornell, env=B.make_env(**env))
    W = H = 128
    cam = _cam(cornell, W, H)
    S = orc.Scene(cornell["flat"])
    want, _, cnt = S.render(_ocam(orc, cam), orc.make_env(**env), W, H, 32, 16, want_counters=True)
    try:
        for g in (2, 1, 0):
            gpu.set_option("groups", g)
            for count in (1, 0):
                gpu.set_option("count", count)
                got, _ = gpu.render(cam, W, H, 32, 16)
                st = gpu.stats()
                gpu.set_option("count", 0)
                assert_bitwise(got, want, "cornell, groups=%d count=%d" % (g, count))
                if count:
                    for k in ("samples", "rays", "scatters", "nan_retries"):
                        assert st[k] == cnt[k], (g, k)
        # every pixel running at once, few of them per wave: the regime the group walk is for (one shard of eight, 7 spp: no sort)
        gpu.set_pixel_shard(5, 8, 16)
        parts = {}
        for g in (2, 1, 0):
            gpu.set_option("groups", g)
            parts[g], _ = gpu.render(cam, W, H, 7, 16)
        gpu.set_pixel_shard(0, 1, 16)
        assert_bitwise(parts[2], parts[0], "shard, forced group walk")
        assert_bitwise(parts[1], parts[0], "shard, sparse-wave group walk")
        # deep tree, full-size frame, every pixel
        _, mats = scene_io.parse_scene(os.path.join(os.path.dirname(B.HEADER_PATH), "..", "assets", "dragon.json"))
        ents = scene_io.build_entities(procedural.dragon_standin(), mats)
        gpu.upload_scene(ents, [m for _, m, _ in mats], env=B.make_env(**env))
        W, H = 1920, 1080
        cam = mkcam([4, 2.5, 0], [0, 0.75, 0], [0, 1, 0], 50, W, H)
        imgs = {}
        for g in (2, 1, 0):
            gpu.set_option("groups", g)
            imgs[g], _ = gpu.render(cam, W, H, 4, 16)
        assert imgs[0].std() > 0.01
        assert_bitwise(imgs[2], imgs[0], "C4 whole frame, forced group walk")
        assert_bitwise(imgs[1], imgs[0], "C4 whole frame, sparse-wave group walk")
    finally:
        gpu.set_option("groups", 1)
        gpu.set_option("count", 0)
        gpu.set_pixel_shard(0, 1, 16)


def test_express_pixels_bitwise(gpu, orc, cornell):
    """Round 3: the most expensive entries of the cost-ordered queue can be rendered as EXPRESS pixels - waves of their own, few pixels
    each, every sample of a pixel in one go, group walk, raised wave priority (pt_kernel.hip take_ticket).  A pixel's stream does not
    depend on who renders it: any share of express pixels and any number of them per wave gives the oracle's image; the other
    waves must finish express pixels that are left over (more express pixels than express slots)."""
    env = dict(color=(1, 1, 1), intensity=0.0)
    _upload(gpu, cornell, env=B.make_env(**env))
    W, H = 96, 72
    cam = _cam(cornell, W, H)
    want, _, cnt = orc.Scene(cornell["flat"]).render(_ocam(orc, cam), orc.make_env(**env), W, H, 40, 16, want_counters=True)
    try:
        gpu.set_option("whole", 0)  # the ring schedule (a tier plan would serve these pixels by cost class instead: test_tier_schedule_bitwise)
        for permille, nse in ((0, 8), (10, 8), (200, 4), (500, 1), (500, 64)):
            gpu.set_option("express_permille", permille)
            gpu.set_option("ns_express", nse)
            gpu.set_option("count", 1)
            got, _ = gpu.render(cam, W, H, 40, 16)
            st = gpu.stats()
            gpu.set_option("count", 0)
            assert_bitwise(got, want, "express_permille=%d ns_express=%d" % (permille, nse))
            assert (st["express_pixels"] > 0) == (permille > 0), (permille, st["express_pixels"])
            for k in ("samples", "rays", "scatters"):
                assert st[k] == cnt[k], (permille, nse, k)
    finally:
        gpu.set_option("express_permille", -1)
        gpu.set_option("ns_express", 8)
        gpu.set_option("count", 0)
        gpu.set_option("whole", -1)


def test_vgpr_fallback_instance(gpu, orc, cube, scene_io):
    """The wavefront kernel's second instance (168-VGPR budget, 12 waves per CU) is what the library launches when the 128-VGPR
    instance of a build would need scratch; option "fallback" forces it.  Same source, same arithmetic: C1 bit for bit."""
    tex = scene_io.checker_texture()
    env = B.make_env(use_auto=True, intensity=1.0)
    _upload(gpu, cube, textures=[tex], mesh_textures=[0], env=env)
    W = H = 256
    cam = _cam(cube, W, H)
    S = orc.Scene(cube["flat"])
    want, _, _ = S.render(_ocam(orc, cam), orc.make_env(use_auto=True, intensity=1.0), W, H, 16, 4)
    gpu.set_option("fallback", 1)
    try:
        got, _ = gpu.render(cam, W, H, 16, 4)
        st = gpu.stats()
    finally:
        gpu.set_option("fallback", 0)
    assert st["kernel_variant"] == 3 and 128 < st["vgprs"] <= 168, (st["kernel_variant"], st["vgprs"])
    assert_bitwise(got, want, "C1 through the fallback instance")
    got, _ = gpu.render(cam, W, H, 16, 4)
    assert gpu.stats()["kernel_variant"] == 2 and gpu.stats()["vgprs"] <= 128


def test_vgpr_fallback_build():
    """A build whose product instance DOES spill (libmi355pt_spilltest.so: -DPT_WAVES_PER_EU=5, 96 VGPRs + scratch): pt_render must
    pick the fallback instance by itself instead of refusing (round 2: PT_E_LIMIT for every user after a compiler bump), and C1 must
    come out bit for bit.  Runs in a child process because the library path is fixed at import."""
    import subprocess, sys
    from conftest import ROOT

    lib = os.path.join(ROOT, "owl-path-tracer_amd", "libmi355pt_spilltest.so")
    if not os.path.exists(lib):
        pytest.skip("libmi355pt_spilltest.so not built (make -C owl-path-tracer_amd/csrc spilltest)")
    code = r"""
import os, sys, numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "oracle"))
import ptamd; ptamd.load()
from owl_path_tracer_amd.pyhost import binding as B, scene_io
import oracle as orc
assert B.LIB_PATH.endswith("libmi355pt_spilltest.so"), B.LIB_PATH
sc = scene_io.load_scene_dir(os.path.join(%(root)r, "assets"), "cube")
tex = scene_io.checker_texture()
ctx = B.Context(0)
ctx.upload_scene(sc["entities"], [m for _, m, _ in sc["materials"]], textures=[tex], mesh_textures=[0], env=B.make_env(use_auto=True, intensity=1.0))
W = H = 256
c = sc["camera"]
cam = B.to_camera_data(c["look_from"], c["look_at"], c["look_up"], c["vertical_fov"], W, H)
got, _ = ctx.render(cam, W, H, 16, 4)
st = ctx.stats()
assert st["kernel_variant"] == 3 and 128 < st["vgprs"] <= 168, (st["kernel_variant"], st["vgprs"])
flat = scene_io.flatten_scene(sc["entities"], sc["materials"], {0: tex})
want, _, _ = orc.Scene(flat).render(orc.to_camera_data(c["look_from"], c["look_at"], c["look_up"], c["vertical_fov"], W, H), orc.make_env(use_auto=True, intensity=1.0), W, H, 16, 4)
assert (got.view(np.uint32) == want.view(np.uint32)).all(), int((got.view(np.uint32) != want.view(np.uint32)).sum())
print("FALLBACK_OK", st["vgprs"])
""" % dict(root=ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PT_LIB_PATH=lib), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "FALLBACK_OK" in r.stdout, r.stdout + r.stderr


def test_lobe_bins_build():
    """Lobe-coherent hit passes (round 4; libmi355pt_lobebins.so - the product build leaves them out, profiles/r04_notes.md): hits are
    binned by the lobe their next draw will pick and a pass shades one bin at a time.  The image must not depend on it: a scene that
    samples all four lobes (the car.json materials on spheres, textured ground, emitter), bins on with every threshold for a pass over
    one bin alone (tune4: 1 = always .. 200 = never), and automatic."""
    import subprocess, sys
    from conftest import ROOT

    lib = os.path.join(ROOT, "owl-path-tracer_amd", "libmi355pt_lobebins.so")
    if not os.path.exists(lib):
        pytest.skip("libmi355pt_lobebins.so not built (make -C owl-path-tracer_amd/csrc lobebins)")
    code = r"""
import os, sys, numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "oracle"))
import ptamd; ptamd.load()
from owl_path_tracer_amd.pyhost import binding as B, scene_io, procedural
import oracle as orc
assert B.LIB_PATH.endswith("libmi355pt_lobebins.so"), B.LIB_PATH
_, car = scene_io.parse_scene(os.path.join(%(root)r, "assets", "car.json"))
mats = [(n, m, "") for n, m, _ in car]
meshes = []
for i, (name, m, _) in enumerate(mats):
    if name == "Ground":
        meshes.append((name, procedural.quad((-6, 0, -6), (-6, 0, 6), (6, 0, 6), (6, 0, -6), (0, 1, 0), uv=True)))
    elif name == "Light":
        meshes.append((name, procedural.quad((-2, 4, -2), (2, 4, -2), (2, 4, 2), (-2, 4, 2), (0, -1, 0))))
    else:
        a = 2 * np.pi * i / len(mats)
        meshes.append((name, procedural.uv_sphere((2.2 * np.cos(a), 0.5, 2.2 * np.sin(a)), 0.5, nu=24, nv=12)))
ents = scene_io.build_entities(meshes, mats)
gi = [n for n, _, _ in mats].index("Ground")
tex = scene_io.checker_texture(32, 32, 4)
env = dict(use_auto=True, intensity=0.6)
ctx = B.Context(0)
ctx.upload_scene(ents, [m for _, m, _ in mats], textures=[tex], mesh_textures=[0 if mid == gi else -1 for _, mid in ents], env=B.make_env(**env))
W, H = 96, 64
cam = B.to_camera_data([0, 3.5, 6.5], [0, 0.4, 0], [0, 1, 0], 45, W, H)
S = orc.Scene(scene_io.flatten_scene(ents, mats, {gi: tex}))
want, _, cnt = S.render(orc.to_camera_data([0, 3.5, 6.5], [0, 0.4, 0], [0, 1, 0], 45, W, H), orc.make_env(**env), W, H, 24, 16, want_counters=True)
ctx.set_option("count", 1); ctx.set_option("lobe_bins", 1)
ctx.render(cam, W, H, 24, 16)
st = ctx.stats(); lb = st["lobes"]
ctx.set_option("count", 0)
assert all(lb[i] > 0 for i in range(5)) and lb[7] > 0, lb   # four lobes + emitter hits; some passes shaded one bin alone
assert st["scatters"] == cnt["scatters"] and st["rays"] == cnt["rays"]
for bins, pure_min in ((0, 0), (-1, 0), (1, 1), (1, 8), (1, 40), (1, 200)):
    ctx.set_option("lobe_bins", bins); ctx.set_option("tune4", pure_min)
    got, _ = ctx.render(cam, W, H, 24, 16)
    bad = int((got.view(np.uint32) != want.view(np.uint32)).sum())
    assert bad == 0, (bins, pure_min, bad)
print("LOBE_BINS_OK")
""" % dict(root=ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PT_LIB_PATH=lib), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "LOBE_BINS_OK" in r.stdout, r.stdout + r.stderr


def test_library_communicator_single_rank(gpu, cornell):
    """The library's own RCCL path on one GPU: unique id, ncclCommInitRank(world 1), pt_render with the reduce in it, pinned
    output - bit-identical to the plain render.  (N > 1 through the same calls: tests/test_host_main.py, bench.py --gpus N.)"""
    _upload(gpu, cornell)
    W, H = 96, 80
    cam = _cam(cornell, W, H)
    want, want8 = gpu.render(cam, W, H, 16, 16, want_rgba8=True)
    ctx = B.Context(0)
    try:
        _upload(ctx, cornell)
        uid = B.comm_unique_id()
        assert len(uid) == 128 and any(uid)
        ctx.comm_init_rank(uid, 0, 1)
        fr = B.PinnedFrame(W, H, want_rgba8=True)
        ctx.render_into(cam, W, H, 16, 16, fr.rgb, fr.rgba8)
        assert_bitwise(np.array(fr.rgb), want, "render with a world-1 communicator")
        np.testing.assert_array_equal(np.array(fr.rgba8), want8)
        ctx.comm_destroy()
        fr.free()
    finally:
        ctx.close()


@pytest.mark.parametrize("key", ["diffuse_roughness(0.0)", "diffuse_roughness(1.0)", "metallic_roughness(0.0)", "specular_transmission_roughness(0.0)", "metallic_vndf_roughness(1.0)"])
def test_furnace_against_reference_rendered_images_gpu(gpu, orc, scene_io, procedural, key):
    """The HIP path against the reference-rendered furnace images (tests/furnace_common.py; tests/test_oracle_render.py holds the
    same check for the oracle) and, bit for bit, against the oracle on the same frame."""
    import furnace_common as F

    ents, mats = F.setup(scene_io, procedural, key)
    gpu.upload_scene(ents, [m for _, m, _ in mats], env=B.make_env(color=(1, 1, 1), intensity=1.0))
    cam = mkcam([3, 1, 0], [0, 1, 0], [0, 1, 0], 50, F.W, F.H)
    rgb, rgba = gpu.render(cam, F.W, F.H, F.SPP, F.DEPTH, want_rgba8=True)
    F.check(key, rgba)
    S = orc.Scene(scene_io.flatten_scene(ents, mats))
    want, want8, _ = S.render(_ocam(orc, cam), orc.make_env(color=(1, 1, 1), intensity=1), F.W, F.H, F.SPP, F.DEPTH, want_rgba8=True)
    assert_bitwise(rgb, want, "furnace " + key)
    np.testing.assert_array_equal(rgba, want8)


@pytest.mark.parametrize("builder", [1, 2])
def test_device_lbvh_builder(gpu, orc, cornell, scene_io, procedural, builder):
    """(builder 2, round 3: PLOC - bottom-up clustering by smallest union area among Morton neighbours, csrc/pt_lbvh.hip; the host lays
    the hierarchy out, pt_bvh_from_hierarchy.)  SURVEY 8(f4): the BVH2 built on the device (option bvh_builder = 1, csrc/pt_lbvh.hip - Morton codes, radix sort, Karras' binary
    radix tree, bottom-up boxes, leaf collapse) replaces owlGroupBuildAccel (application.cpp:131-140).  Closest hit does not depend on
    the tree, so: 20 000 closest-hit queries bit-exact against the oracle's brute force, the product tree walked on the host agrees,
    and the image equals the oracle's bit for bit - for the cornell box (17 974 triangles) and for a 12-material scene; every leaf
    size; structure checks (every triangle in exactly one leaf, leaf sizes, depth bound)."""
    ctx = B.Context(0)
    try:
        for leaf in (1, 4, 7):
            ctx.set_option("bvh_builder", builder)
            ctx.set_option("leaf_size", leaf)
            _upload(ctx, cornell)
            st = ctx.stats()
            assert 0 < st["bvh_nodes"] < 17974 and 0 < st["bvh_depth"] <= 64, st
            oi = ctx.oct_info()  # every triangle in exactly one leaf and inside its leaf's box (pt_debug_oct_info fails otherwise)
            assert oi["triangles"] == 17974
            S = orc.Scene(cornell["flat"])
            rng = np.random.default_rng(11 + leaf)
            n = 20000 if leaf == 4 else 4000
            o = rng.uniform(-1.0, 1.0, (n, 3)).astype(np.float32) + np.array([0, 1, 0], np.float32)
            d = rng.normal(size=(n, 3)).astype(np.float32)
            d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
            got = ctx.debug_eval("closest_hit", np.concatenate([o, d], 1), 5)
            for i in range(0, n, 97):
                hit, t, u, v, prim = S.intersect(o[i], d[i], use_bvh=False)
                assert bool(got[i, 0]) == hit
                if hit:
                    assert_bitwise(got[i, 1:4], np.float32([t, u, v]), "LBVH leaf %d hit %d" % (leaf, i))
                    assert int(got[i, 4:5].view(np.int32)[0]) == prim, (leaf, i)
                    hh, ht, hu, hv, hp = ctx.closest_hit_host(o[i], d[i])
                    assert hh and hp == prim
                    assert_bitwise(np.float32([ht, hu, hv]), np.float32([t, u, v]), "LBVH walked on the host")
        ctx.set_option("leaf_size", 4)
        _upload(ctx, cornell)
        W, H = 160, 120
        cam = _cam(cornell, W, H)
        got, _ = ctx.render(cam, W, H, 32, 16)
        want, _, _ = orc.Scene(cornell["flat"]).render(_ocam(orc, cam), orc.make_env(color=(1, 1, 1), intensity=0.0), W, H, 32, 16)
        assert_bitwise(got, want, "cornell image through the device-built BVH")
        # same image as the host-built tree on the same context type
        _upload(gpu, cornell)
        ref, _ = gpu.render(cam, W, H, 32, 16)
        assert_bitwise(got, ref, "device-built vs host-built BVH")
        print("device builder %d, cornell: %d nodes depth %d build %.2f ms" % (builder, ctx.stats()["bvh_nodes"], ctx.stats()["bvh_depth"], ctx.stats()["bvh_build_ms"]))
    finally:
        ctx.close()


def test_group_scene_replica_is_a_clone(gpu, cube, scene_io):
    """pt_group_upload_scene builds the BVH once and clones the scene into the other devices' contexts (pti::clone_scene): a cloned
    context renders the same image and accepts a material hot-swap like the original (two contexts on device 0)."""
    tex = scene_io.checker_texture()
    _upload(gpu, cube, textures=[tex], mesh_textures=[0], env=B.make_env(use_auto=True, intensity=1.0))
    W, H = 128, 96
    cam = _cam(cube, W, H)
    want, want8 = gpu.render(cam, W, H, 16, 4, want_rgba8=True)
    other = B.Context(0)
    try:
        other.clone_scene_from(gpu)
        got, got8 = other.render(cam, W, H, 16, 4, want_rgba8=True)
        assert_bitwise(got, want, "cloned scene")
        np.testing.assert_array_equal(got8, want8)
        assert other.stats()["bvh_nodes"] == gpu.stats()["bvh_nodes"] and other.quad_info() == gpu.quad_info()
        mats = np.stack([m for _, m, _ in cube["materials"]]).copy()
        mats[0, 4] = 0.9
        gpu.set_materials(mats); other.set_materials(mats)
        a, _ = gpu.render(cam, W, H, 16, 4); b, _ = other.render(cam, W, H, 16, 4)
        assert_bitwise(a, b, "cloned scene after pt_set_materials")
        assert not (a.view(np.uint32) == want.view(np.uint32)).all()
    finally:
        other.close()
