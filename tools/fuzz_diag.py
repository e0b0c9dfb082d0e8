"""Run ONE fuzz case (tests/test_gpu_fuzz.py, `python tools/fuzz_diag.py <seed> [large]`) through every render path of the library and say
which ones differ from the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import ptamd
ptamd.load()
import oracle as orc
from owl_path_tracer_amd.pyhost import scene_io, binding as B
import test_gpu_fuzz as F

seed, large = int(sys.argv[1]), len(sys.argv) > 2
gpu = B.Context(0)
for path in F.PATHS:
    if path and path[0][0] == "shard":
        continue
    try:
        F._case(gpu, orc, scene_io, seed, large, path=path)
        print("path %-70s identical" % (path,))
    except AssertionError as e:
        print("path %-70s DIFFERS: %s" % (path, str(e).split("): ")[-1][:160]))
