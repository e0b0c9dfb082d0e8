"""Accuracy of the deterministic libm (dm_*) against float64.  The bounds are the contract quoted in
DESIGN.md; CUDA's libdevice (what the reference ran on) documents sinf/cosf 2, tanf 4, atanf 2, asinf 2,
logf 1, expf 2, powf 4 ulp -- the same order."""
import numpy as np

from conftest import ulp_err

RNG = np.random.default_rng(20240917)
N = 400_000


def test_sin_cos_tan(orc):
    x = RNG.uniform(-8, 8, N).astype(np.float32)
    x64 = x.astype(np.float64)
    assert ulp_err(orc.dm("sin", x), np.sin(x64)).max() <= 2.0
    assert ulp_err(orc.dm("cos", x), np.cos(x64)).max() <= 2.0
    assert ulp_err(orc.dm("tan", x), np.tan(x64)).max() <= 4.0
    # exact points used by the path
    assert orc.dm("sin", np.float32(0))[0] == 0.0 and orc.dm("cos", np.float32(0))[0] == 1.0


def test_atan_asin_atan2(orc):
    x = np.concatenate([RNG.uniform(-50, 50, N), RNG.standard_cauchy(N) * 100]).astype(np.float32)
    assert ulp_err(orc.dm("atan", x), np.arctan(x.astype(np.float64))).max() <= 3.0
    assert orc.dm("atan", np.float32(np.inf))[0] == np.float32(np.pi / 2)
    x = RNG.uniform(-1, 1, N).astype(np.float32)
    assert ulp_err(orc.dm("asin", x), np.arcsin(x.astype(np.float64))).max() <= 3.0
    y = RNG.uniform(-3, 3, N).astype(np.float32)
    x = RNG.uniform(-3, 3, N).astype(np.float32)
    assert ulp_err(orc.dm("atan2", y, x), np.arctan2(y.astype(np.float64), x.astype(np.float64))).max() <= 3.5


def test_log_exp_pow(orc):
    x = np.exp(RNG.uniform(-30, 30, N)).astype(np.float32)
    assert ulp_err(orc.dm("log", x), np.log(x.astype(np.float64))).max() <= 1.0
    x = RNG.uniform(-80, 80, N).astype(np.float32)
    assert ulp_err(orc.dm("exp", x), np.exp(x.astype(np.float64))).max() <= 1.5
    # the two call sites: powf(c, 2.2f) (disney_helper.cuh:6) and powf(alpha2, 1-u) (disney_clearcoat.cuh:26)
    c = RNG.uniform(0, 1, N).astype(np.float32)
    assert ulp_err(orc.dm("pow", c, np.float32(2.2)), np.power(c.astype(np.float64), np.float64(np.float32(2.2)))).max() <= 2.0
    a2 = (RNG.uniform(0.001, 0.1, N) ** 2).astype(np.float32)
    u = RNG.uniform(0, 1, N).astype(np.float32)
    assert ulp_err(orc.dm("pow", a2, u), np.power(a2.astype(np.float64), u.astype(np.float64))).max() <= 2.0
    got = orc.dm("pow", np.array([0, 1, 0.5, 2, 4], np.float32), np.array([2.2, 5, 2, 10, 0.5], np.float32))
    np.testing.assert_array_equal(got, np.array([0, 1, 0.25, 1024, 2], np.float32))


def test_sqrt_div_correctly_rounded(orc):
    x = np.exp(RNG.uniform(-30, 30, N)).astype(np.float32)
    y = RNG.uniform(-3, 3, N).astype(np.float32)
    np.testing.assert_array_equal(orc.dm("sqrt", x), np.sqrt(x))
    np.testing.assert_array_equal(orc.dm("div", x, y), x / y)
