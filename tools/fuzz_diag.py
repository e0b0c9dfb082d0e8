"""Run ONE fuzz case (tests/test_gpu_fuzz.py) through every render path of the library and say which ones differ from the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import ptamd
ptamd.load()
import oracle as orc
from owl_path_tracer_amd.pyhost import scene_io, binding as B
import test_gpu_fuzz as F

seed = int(sys.argv[1])
gpu = B.Context(0)
for path in F.PATHS:
    F.PATHS_SAVE = F.PATHS
    class R:  # force the path choice: wrap the rng used for the path draw by replacing PATHS with a one-element list
        pass
    F.PATHS = [path] * len(F.PATHS_SAVE)
    try:
        F._case(gpu, orc, scene_io, seed)
        print("path %-60s identical" % (path,))
    except AssertionError as e:
        print("path %-60s DIFFERS: %s" % (path, str(e).split("): ")[1][:160]))
    F.PATHS = F.PATHS_SAVE
