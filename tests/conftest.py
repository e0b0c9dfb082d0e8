import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import ptamd  # noqa: E402

ptamd.load()

ASSETS = os.path.join(ROOT, "assets")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    import oracle

    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def scene_io():
    from owl_path_tracer_amd.pyhost import scene_io as m

    return m


@pytest.fixture(scope="session")
def procedural():
    from owl_path_tracer_amd.pyhost import procedural as m

    return m


@pytest.fixture(scope="session")
def cornell(scene_io):
    sc = scene_io.load_scene_dir(ASSETS, "cornell-box")
    sc["flat"] = scene_io.flatten_scene(sc["entities"], sc["materials"])
    return sc


@pytest.fixture(scope="session")
def cube(scene_io):
    sc = scene_io.load_scene_dir(ASSETS, "cube")
    # cube.json is textured; the PNG was never committed upstream -> deterministic checker stand-in
    sc["flat"] = scene_io.flatten_scene(sc["entities"], sc["materials"], {0: scene_io.checker_texture()})
    return sc


def ulp_err(got, ref64):
    ref32 = ref64.astype(np.float32)
    ulp = np.spacing(np.abs(ref32)).astype(np.float64)
    ulp = np.maximum(ulp, np.float64(np.finfo(np.float32).tiny))
    return np.abs(got.astype(np.float64) - ref64) / ulp
