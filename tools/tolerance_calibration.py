#!/usr/bin/env python3
"""CPU-vs-CPU noise floor for the stated OptiX tolerance: the oracle as shipped (deterministic dm_* math,
-ffp-contract=off) against the SAME source built with the host libm and -ffp-contract=fast (oracle/libpt_oracle_hostlibm.so) - two
builds that differ the way an independent toolchain would (other transcendentals, other fma choices), same RNG streams.

  python tools/tolerance_calibration.py [W H spp]      -> one JSON line with the metrics of SURVEY 8(d) "Parity statement"
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "oracle"))
import ptamd; ptamd.load()
from owl_path_tracer_amd.pyhost import scene_io
import oracle as orc
sc = scene_io.load_scene_dir(os.path.join(%(root)r, "assets"), "cornell-box")
S = orc.Scene(scene_io.flatten_scene(sc["entities"], sc["materials"]))
c = sc["camera"]
W, H, spp = %(W)d, %(H)d, %(spp)d
cam = orc.to_camera_data(c["look_from"], c["look_at"], c["look_up"], c["vertical_fov"], W, H)
rgb, rgba, _ = S.render(cam, orc.make_env(color=(1, 1, 1), intensity=0.0), W, H, spp, 16, want_rgba8=True)
np.save(%(out)r + "_rgb.npy", rgb); np.save(%(out)r + "_rgba.npy", rgba)
'''


def render(lib, W, H, spp, out):
    env = dict(os.environ)
    if lib:
        env["PT_ORACLE_LIB"] = lib
    subprocess.check_call([sys.executable, "-c", CHILD % dict(root=ROOT, W=W, H=H, spp=spp, out=out)], env=env)
    return np.load(out + "_rgb.npy"), np.load(out + "_rgba.npy")


def metrics(a, a8, b, b8):
    d = np.sqrt(((a.astype(np.float64) - b.astype(np.float64)) ** 2).sum(-1))
    lum = (0.2126 * a[..., 0] + 0.7152 * a[..., 1] + 0.0722 * a[..., 2]).astype(np.float64)
    rmse = float(np.sqrt(((a.astype(np.float64) - b.astype(np.float64)) ** 2).mean()))
    ch = lambda x, k: ((x >> (8 * k)) & 0xFF).astype(np.int32)
    off = np.zeros(a8.shape, bool)
    for k in range(3):
        off |= np.abs(ch(a8, k) - ch(b8, k)) > 1
    return {"pixels": int(d.size), "identical_pixels_pct": round(100.0 * float((d == 0).mean()), 3), "l2_max": float(d.max()),
            "l2_p999": float(np.percentile(d, 99.9)), "l2_p99": float(np.percentile(d, 99.0)), "l2_median": float(np.median(d)),
            "rel_rmse": rmse / float(lum.mean()), "rgba8_off_by_more_than_1_pct": round(100.0 * float(off.mean()), 4)}


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    spp = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    tmp = os.environ.get("TMPDIR", "/tmp")
    a, a8 = render(None, W, H, spp, os.path.join(tmp, "tolcal_det"))
    b, b8 = render(os.path.join(ROOT, "oracle", "libpt_oracle_hostlibm.so"), W, H, spp, os.path.join(tmp, "tolcal_host"))
    m = metrics(a, a8, b, b8)
    m.update(scene="cornell-box (C2)", size=[W, H], spp=spp, depth=16, pair="dm_* math, -ffp-contract=off  vs  glibc libm, -ffp-contract=fast")
    print(json.dumps(m))


if __name__ == "__main__":
    main()
