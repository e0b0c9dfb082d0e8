"""The stated per-pixel tolerance against an independent toolchain, calibrated CPU-vs-CPU as BASELINE.md section 4 prescribes: the
oracle as shipped vs the same source with the host libm and -ffp-contract=fast (tools/tolerance_calibration.py).  Same RNG streams:
the bulk of the pixels agrees to ~1e-7, a fraction of a percent takes a different branch somewhere (an ulp at a lobe threshold, a
Russian-roulette draw, a triangle edge) and moves by one path's worth of Monte-Carlo noise.  The full-size numbers (C2, 512x512,
256 spp) are in DESIGN.md section 2 / profiles/r02_tolerance_calibration.json; this test keeps the bar from silently rotting."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_cpu_vs_cpu_noise_floor_is_within_the_stated_tolerance(tmp_path, monkeypatch):
    import tolerance_calibration as T

    monkeypatch.setenv("TMPDIR", str(tmp_path))
    a, a8 = T.render(None, 256, 256, 256, str(tmp_path / "det"))
    b, b8 = T.render(os.path.join(ROOT, "oracle", "libpt_oracle_hostlibm.so"), 256, 256, 256, str(tmp_path / "host"))
    m = T.metrics(a, a8, b, b8)
    assert m["identical_pixels_pct"] < 100.0  # the two builds really differ
    assert m["l2_p99"] <= 1e-4          # 99 % of the pixels: float noise only
    assert m["l2_p999"] <= 0.04         # the stated bar (calibrated on C2 512x512x256spp: 0.029; 256x256: 0.016)
    assert m["rel_rmse"] <= 1e-2        # the stated bar (calibrated: 5.8e-3)
    assert m["rgba8_off_by_more_than_1_pct"] <= 0.5
