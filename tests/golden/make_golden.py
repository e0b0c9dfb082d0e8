"""Regenerates the self-made regression fixtures under tests/golden/ from the CPU oracle.

These are NOT reference outputs (the reference cannot run here; it ships no fixtures): they pin the
oracle against accidental change and give the GPU suite a committed image to compare with on a box
where the oracle .so is also present.  Run: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import ptamd  # noqa: E402

ptamd.load()
import oracle as orc  # noqa: E402
from owl_path_tracer_amd.pyhost import scene_io  # noqa: E402


def main():
    assets = os.path.join(ROOT, "assets")
    sc = scene_io.load_scene_dir(assets, "cube")
    flat = scene_io.flatten_scene(sc["entities"], sc["materials"], {0: scene_io.checker_texture()})
    S = orc.Scene(flat)
    c = sc["camera"]
    cam = orc.to_camera_data(c["look_from"], c["look_at"], c["look_up"], c["vertical_fov"], 32, 32)
    rgb, _, _ = S.render(cam, orc.make_env(use_auto=True, intensity=1), 32, 32, 8, 4)
    np.save(os.path.join(HERE, "cube_32x32_8spp_d4_oracle.npy"), rgb)

    sc = scene_io.load_scene_dir(assets, "cornell-box")
    flat = scene_io.flatten_scene(sc["entities"], sc["materials"])
    S = orc.Scene(flat)
    c = sc["camera"]
    cam = orc.to_camera_data(c["look_from"], c["look_at"], c["look_up"], c["vertical_fov"], 48, 48)
    rgb, _, _ = S.render(cam, orc.make_env(color=(1, 1, 1), intensity=0), 48, 48, 16, 16)
    np.save(os.path.join(HERE, "cornell_48x48_16spp_d16_oracle.npy"), rgb)


if __name__ == "__main__":
    main()
