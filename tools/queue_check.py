#!/usr/bin/env python3
"""Dump the cost-ordered queue of a C4 render: python tools/queue_check.py [spp]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ptamd
ptamd.load()
from owl_path_tracer_amd.pyhost import binding as B, scene_io, procedural
ctx = B.Context(0)
_, mats = scene_io.parse_scene(os.path.join(ROOT, "assets", "dragon.json"))
ents = scene_io.build_entities(procedural.dragon_standin(4357, 100), mats)
W, H, spp = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 32
cam = B.to_camera_data([4, 2.5, 0], [0, .75, 0], [0, 1, 0], 50, W, H)
ctx.upload_scene(ents, [m for _, m, _ in mats], env=B.make_env(color=(1, 1, 1), intensity=0.0))
ctx.render(cam, W, H, spp, 16)
q, i, c = ctx.read_queue(W * H)
print("n", len(q), "launches", ctx.stats()["launches"])
print("cost hist", np.bincount(c, minlength=64)[:64].tolist())
cost_of = np.zeros(W * H, np.uint8); cost_of[i] = c
cq = cost_of[q]
print("permutation ok", np.array_equal(np.sort(q), np.sort(i)))
print("cost along queue (mean per 1/20th):", [round(float(x.mean()), 2) for x in np.array_split(cq, 20)])
