#!/usr/bin/env python3
"""node / triangle visits per ray of the per-lane quad walk (groups=0) and the group walk over oct nodes (groups=2), same frame"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ptamd
ptamd.load()
from owl_path_tracer_amd.pyhost import binding as B, scene_io, procedural
which = sys.argv[1] if len(sys.argv) > 1 else "c4"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = B.Context(0)
if which == "c4":
    _, mats = scene_io.parse_scene(os.path.join(ROOT, "assets", "dragon.json"))
    ents = scene_io.build_entities(procedural.dragon_standin(), mats)
    W, H = 1920, 1080
    cam = B.to_camera_data([4, 2.5, 0], [0, .75, 0], [0, 1, 0], 50, W, H)
    ctx.upload_scene(ents, [m for _, m, _ in mats], env=B.make_env(color=(1, 1, 1), intensity=0.0))
else:
    sc = scene_io.load_scene_dir(os.path.join(ROOT, "assets"), "cornell-box")
    W = H = 512
    c = sc["camera"]
    cam = B.to_camera_data(c["look_from"], c["look_at"], c["look_up"], c["vertical_fov"], W, H)
    ctx.upload_scene(sc["entities"], [m for _, m, _ in sc["materials"]], env=B.make_env(color=(1, 1, 1), intensity=0.0))
for g in (0, 2):
    ctx.set_option("groups", g)
    ctx.set_option("count", 1)
    ctx.render(cam, W, H, spp, 16)
    st = ctx.stats()
    ctx.set_option("count", 0)
    ctx.render(cam, W, H, spp, 16)
    ms = ctx.stats()["kernel_ms"]
    unit = 4 if g == 2 else 2
    print(json.dumps({"scene": which, "groups": g, "rays": st["rays"], "node_visits_per_ray": round(st["nodes"] / unit / st["rays"], 2), "tris_per_ray": round(st["tris"] / st["rays"], 2),
                      "group_iters_per_ray": round(st["groups"][2] / max(1, st["groups"][5]), 2), "kernel_ms": round(ms, 2)}))
