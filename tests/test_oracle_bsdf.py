"""Behavioural checks of the oracle's restatement of the Disney BSDF and frame math, including the
bug-for-bug quirks listed in SURVEY.md appendix B.  No reference fixtures exist for these (parity
unpinned); the checks are against the formulas in the reference source (cited per test)."""
import numpy as np
import pytest

LOBE_NONE, LOBE_DIFFUSE, LOBE_CLEARCOAT, LOBE_METALLIC, LOBE_GLASS = -1, 0, 1, 2, 3


def _unit(v):
    v = np.asarray(v, np.float64)
    return (v / np.linalg.norm(v)).astype(np.float32)


def test_frame_math(orc):
    # math.hpp:86-107: onb picks (1,1,1)xN unless n.x==n.y==n.z; to_local/to_world normalise
    for n in ([0, 1, 0], [0, 0, 1], [1, 0, 0], _unit([1, 2, 3]), _unit([1, 1, 1]), _unit([-1, -1, -1])):
        n = np.asarray(n, np.float32)
        t, b = orc.onb(n)
        assert abs(np.dot(t, n)) < 1e-6 and abs(np.dot(b, n)) < 1e-6 and abs(np.dot(t, b)) < 1e-6
        assert abs(np.linalg.norm(t) - 1) < 1e-6
        w = _unit([0.3, -0.5, 0.8])
        lw = orc.to_local(t, b, n, w)
        np.testing.assert_allclose(orc.to_world(t, b, n, lw), w, atol=2e-6)
        # quirk 19: results are normalised even for non-unit input
        np.testing.assert_allclose(np.linalg.norm(orc.to_local(t, b, n, 3.0 * w)), 1.0, atol=1e-6)


def test_cosine_hemisphere(orc):
    # sample_methods.hpp:19-41,53-60: concentric disk; wi.z >= 0; (0.5,0.5) -> pole
    np.testing.assert_array_equal(orc.sample_cosine_hemisphere(0.5, 0.5), [0, 0, 1])
    rng = np.random.default_rng(3)
    for u0, u1 in rng.uniform(0, 1, (200, 2)):
        w = orc.sample_cosine_hemisphere(float(u0), float(u1))
        assert w[2] >= 0 and abs(np.linalg.norm(w) - 1) < 1e-6
        dx, dy = 2 * np.float32(u0) - 1, 2 * np.float32(u1) - 1
        if abs(dx) > abs(dy):
            r, phi = dx, (np.pi / 4) * (dy / dx)
        else:
            r, phi = dy, np.pi / 2 - (np.pi / 4) * (dx / dy)
        np.testing.assert_allclose(w[:2], [r * np.cos(phi), r * np.sin(phi)], atol=2e-6)


def test_refract_and_fresnel(orc):
    # math.hpp:63-77
    ok, wi = orc.refract([0, 0, 1], [0, 0, 1], 1.0)
    assert ok and list(wi) == [0, 0, -1]
    w = _unit([0.6, 0, 0.8])
    ok, wi = orc.refract(w, [0, 0, 1], 1 / 1.5)
    assert ok
    np.testing.assert_allclose(np.hypot(wi[0], wi[1]), 0.6 / 1.5, atol=1e-6)  # Snell
    ok, _ = orc.refract(_unit([0.9, 0, 0.2]), [0, 0, 1], 1.5)  # TIR
    assert not ok
    # disney_helper.cuh:52-60: normal incidence -> ((n-1)/(n+1))^2 ; TIR -> 1
    assert abs(orc.fresnel_equation([0, 0, 1], [0, 0, 1], 1.0, 1.5) - 0.04) < 1e-6
    assert orc.fresnel_equation(_unit([0.9, 0, 0.2]), [0, 0, 1], 1.5, 1.0) == 1.0


def test_ndf_and_masking(orc):
    # disney_specular.cuh:54-60: isotropic GGX at the pole = 1/(pi a^2); :17-27 lambda(pole)=0
    a = 0.3
    assert abs(orc.d_gtr2([0, 0, 1], a, a) - 1 / (np.pi * a * a)) < 1e-3
    assert orc.lambda_([0, 0, 1], a, a) == 0.0
    assert orc.d_gtr2([1, 0, 0], a, a) == 0.0  # tan^2 = inf
    w = _unit([0.5, 0.2, 0.6])
    tan2 = (1 - w[2] ** 2) / w[2] ** 2
    assert abs(orc.lambda_(w, a, a) - (-1 + np.sqrt(1 + a * a * tan2)) / 2) < 1e-6
    # disney_clearcoat.cuh:13-20
    assert abs(orc.d_gtr1([0, 0, 1], 1.0) - 1 / np.pi) < 1e-7
    a2 = 0.05 ** 2
    assert abs(orc.d_gtr1([0, 0, 1], 0.05) - (a2 - 1) / (np.pi * np.log(a2) * a2)) / orc.d_gtr1([0, 0, 1], 0.05) < 1e-5


def test_lobe_selection_order_and_draw_counts(orc, scene_io):
    # disney.cuh:44-63: thresholds metallic -> clearcoat -> diffuse -> glass, '<=' comparisons
    m = scene_io.material(metallic=0.25, clearcoat=1.0, specular_transmission=0.5, roughness=0.5, clearcoat_gloss=0.5)
    w_d, w_m, w_c, w_g = (1 - 0.5) * (1 - 0.25), 0.25, 0.25, (1 - 0.25) * 0.5
    tot = w_d + w_m + w_c + w_g
    wo = _unit([0.2, 0.1, 0.9])
    seen = set()
    for seed in range(400):
        st = orc.rng_init(seed, 1)
        p, st1 = orc.rng_next(st)
        r = orc.sample_disney(m, wo, st)
        if p <= w_m / tot:
            want = LOBE_METALLIC
        elif p <= (w_m + w_c) / tot:
            want = LOBE_CLEARCOAT
        elif p <= (w_m + w_c + w_d) / tot:
            want = LOBE_DIFFUSE
        else:
            want = LOBE_GLASS
        assert r["lobe"] == want
        seen.add(want)
        # draws: 1 (lobe) + 2 for brdf/clearcoat/diffuse; glass 2 + {0|1} + {0|2}
        s = st1
        for _ in range(2):
            _, s = orc.rng_next(s)
        if want != LOBE_GLASS:
            assert r["state"] == s
        else:
            cands = [s]
            _, s3 = orc.rng_next(s)
            cands.append(s3)
            s5 = s
            for _ in range(2):
                _, s5 = orc.rng_next(s5)
            cands.append(s5)  # refract failed: no 3rd draw, resample (2)
            s6 = s3
            for _ in range(2):
                _, s6 = orc.rng_next(s6)
            cands.append(s6)
            assert r["state"] in cands
    assert seen == {LOBE_METALLIC, LOBE_CLEARCOAT, LOBE_DIFFUSE, LOBE_GLASS}


def test_force_btdf(orc, scene_io):
    # disney.cuh:40: inside (wo.z<0) after a GLASS sample every lobe draw goes to the glass lobe
    m = scene_io.material(metallic=0.5, specular_transmission=0.5)
    wo = _unit([0.1, 0.2, -0.9])
    for seed in range(50):
        assert orc.sample_disney(m, wo, orc.rng_init(seed, 9), LOBE_GLASS)["lobe"] == LOBE_GLASS
    lobes = {orc.sample_disney(m, wo, orc.rng_init(seed, 9), LOBE_DIFFUSE)["lobe"] for seed in range(200)}
    assert LOBE_METALLIC in lobes


def test_diffuse_values(orc, scene_io):
    # disney_diffuse.cuh:26-55
    m = scene_io.material(base_color=[0.2, 0.5, 0.9], roughness=0.7)
    wo, wi = _unit([0.3, 0.1, 0.8]), _unit([-0.2, 0.4, 0.6])
    f, pdf = orc.eval_lobe(LOBE_DIFFUSE, m, wo, [0, 0, 0], wi)
    sw = lambda c: np.clip(1 - c, 0, 1) ** 5
    fo, fi = sw(wo[2]), sw(wi[2])
    fd = (1 - 0.5 * fo) * (1 - 0.5 * fi)
    rr = 0.7 * (np.dot(wo, wi) + 1)
    fr = rr * (fi + fo + fo * fi * (rr - 1))
    np.testing.assert_allclose(f, np.array([0.2, 0.5, 0.9]) / np.pi * (fd + fr), rtol=2e-6)
    assert abs(pdf - abs(wi[2]) / np.pi) < 1e-7


def test_specular_brdf_quirks(orc, scene_io):
    # disney_specular.cuh:125-149: f = D*G*F/(4|wo.z|) (no 1/cos_i), pdf = VNDF pdf D*G1(wo)*max(0,wo.wh)/(4 wo.z)
    m = scene_io.material(base_color=[0.9, 0.6, 0.3], metallic=1.0, roughness=0.4)
    wo, wi = _unit([0.3, 0.1, 0.8]), _unit([-0.25, 0.05, 0.7])
    wh = _unit(wo.astype(np.float64) + wi)
    f, pdf = orc.eval_lobe(LOBE_METALLIC, m, wo, wh, wi)
    a = max(1e-3, 0.4 ** 2)
    D = orc.d_gtr2(wh, a, a)
    lam = lambda w: orc.lambda_(w, a, a)
    G = 1 / (1 + lam(wo) + lam(wi))
    F = np.array([0.9, 0.6, 0.3]) + (1 - np.array([0.9, 0.6, 0.3])) * np.clip(1 - np.dot(wi, wh), 0, 1) ** 5
    np.testing.assert_allclose(f, D * G * F / (4 * abs(wo[2])), rtol=5e-6)
    np.testing.assert_allclose(pdf, D / (1 + lam(wo)) * max(0, np.dot(wo, wh)) / (4 * wo[2]), rtol=5e-6)
    # wo.z < 0 -> negative pdf (path dies at device.cu:193)
    wo2 = _unit([0.3, 0.1, -0.8])
    r = orc.sample_disney(m, wo2, orc.rng_init(3, 3))
    assert r["lobe"] == LOBE_METALLIC and r["pdf"] <= 0


def test_clearcoat_quirks(orc, scene_io):
    # disney_clearcoat.cuh:45-59: F = lerp(1, schlick(wi.z), 0.04) (argument order bug-for-bug); pdf = D/(4 wh.wi)
    m = scene_io.material(clearcoat=1.0, clearcoat_gloss=0.5)
    wo, wi = _unit([0.3, 0.1, 0.8]), _unit([-0.25, 0.05, 0.7])
    wh = _unit(wo.astype(np.float64) + wi)
    f, pdf = orc.eval_lobe(LOBE_CLEARCOAT, m, wo, wh, wi)
    alpha = 0.1 + (0.001 - 0.1) * 0.5
    D = orc.d_gtr1(wh, alpha)
    F = 1 + (np.clip(1 - wi[2], 0, 1) ** 5 - 1) * 0.04
    G = 1 / (1 + orc.lambda_(wo, 0.25, 0.25)) / (1 + orc.lambda_(wi, 0.25, 0.25))
    np.testing.assert_allclose(f, D * G * F / (4 * abs(wo[2]) * abs(wi[2])), rtol=5e-6)
    np.testing.assert_allclose(pdf, D / (4 * np.dot(wh, wi)), rtol=5e-6)
    f0, pdf0 = orc.eval_lobe(LOBE_CLEARCOAT, scene_io.material(clearcoat=0.0), wo, wh, wi)
    assert pdf0 == 0 and not f0.any()


def test_glass_eval(orc, scene_io):
    # disney_specular.cuh:193-214: reflect: pdf=R, f=base*R/|wi.z|; transmit: pdf=T, f=sqrt(base)*T/|wi.z|/eta^2
    base = np.array([0.64, 0.81, 0.25], np.float32)
    m = scene_io.material(base_color=base, specular_transmission=1.0, ior=1.5)
    wo, wh = _unit([0.3, 0.1, 0.8]), np.array([0, 0, 1], np.float32)
    R = orc.fresnel_equation(wo, wh, 1.0, 1.5)
    wi_r = np.array([-wo[0], -wo[1], wo[2]], np.float32)
    f, pdf = orc.eval_lobe(LOBE_GLASS, m, wo, wh, wi_r)
    np.testing.assert_allclose(pdf, R, rtol=1e-6)
    np.testing.assert_allclose(f, base * R / abs(wi_r[2]), rtol=2e-6)
    ok, wi_t = orc.refract(wo, wh, 1 / 1.5)
    f, pdf = orc.eval_lobe(LOBE_GLASS, m, wo, wh, wi_t)
    np.testing.assert_allclose(pdf, 1 - R, rtol=1e-6)
    np.testing.assert_allclose(f, np.sqrt(base) * (1 - R) / abs(wi_t[2]) / (1 / 1.5) ** 2, rtol=3e-6)


def test_sheen(orc, scene_io):
    # disney_sheen.cuh:15-37: tint uses luminance of base^2.2 but the un-linearised base colour
    base = np.array([0.8, 0.4, 0.2])
    m = scene_io.material(base_color=base, sheen=0.7, sheen_tint=0.6)
    wo, wi = _unit([0.3, 0.1, 0.8]), _unit([-0.7, 0.05, 0.3])
    wh = _unit(wo.astype(np.float64) + wi)
    lum = np.dot([0.2126, 0.7152, 0.0722], base ** 2.2)
    want = (1 + (base / lum - 1) * 0.6) * 0.7 * np.clip(1 - np.dot(wi, wh), 0, 1) ** 5
    np.testing.assert_allclose(orc.eval_sheen(m, wo, wi), want, rtol=1e-5)
    assert not orc.eval_sheen(scene_io.material(sheen=0.0), wo, wi).any()
    assert not orc.eval_sheen(m, wo, -wo).any()  # degenerate half vector


def test_sample_disney_returns_lobe_pdf_not_mixture(orc, scene_io):
    # quirk 11 (disney.cuh:65): pdf is the selected lobe's pdf, f the lobe's value (+sheen)
    m = scene_io.material(metallic=0.5, roughness=0.6)
    wo = _unit([0.2, 0.3, 0.85])
    for seed in range(40):
        r = orc.sample_disney(m, wo, orc.rng_init(seed, 5))
        if r["lobe"] == LOBE_DIFFUSE:
            f, pdf = orc.eval_lobe(LOBE_DIFFUSE, m, wo, [0, 0, 0], r["wi"])
            np.testing.assert_array_equal(f, r["f"])
            assert pdf == r["pdf"]


def test_make_rgba_and_texture(orc):
    # owl::make_rgba assumption: min(255, max(0, int(f*256))) packed r | g<<8 | b<<16 | 0xff<<24
    assert orc.make_rgba([0, 0.5, 1.0]) == (0 | (128 << 8) | (255 << 16) | (0xFF << 24))
    assert orc.make_rgba([-1, 0.999, 7.0]) == (0 | (255 << 8) | (255 << 16) | (0xFF << 24))
    assert orc.make_rgba([0.00389, float("nan"), 0.0039063]) & 0xFFFFFF == (0 | (0 << 8) | (1 << 16))
    tex = np.array([[0xFF0000FF, 0xFF00FF00], [0xFFFF0000, 0xFFFFFFFF]], np.uint32)
    np.testing.assert_array_equal(orc.tex_nearest(tex, 0.1, 0.1), [1, 0, 0])
    np.testing.assert_array_equal(orc.tex_nearest(tex, 0.9, 0.1), [0, 1, 0])
    np.testing.assert_array_equal(orc.tex_nearest(tex, 0.1, 0.9), [0, 0, 1])
    np.testing.assert_array_equal(orc.tex_nearest(tex, 1.7, -3.0), [0, 1, 0])  # clamp
    np.testing.assert_allclose(orc.uv_on_sphere([0, 0, 1]), [0.5, 0.5], atol=1e-7)  # device.cu:23-28
    np.testing.assert_allclose(orc.uv_on_sphere([1, 0, 0]), [0.75, 0.5], atol=1e-7)
    np.testing.assert_allclose(orc.uv_on_sphere([0, 1, 0]), [0.5, 1.0], atol=1e-7)
