#!/usr/bin/env python3
"""A/B timing of libmi355pt.so variants on one GPU: PT_LIB_PATH=<.so> python tools/ab_bench.py [c4|c2] [reps] [opt=val ...]"""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ptamd
ptamd.load()
from owl_path_tracer_amd.pyhost import binding as B, scene_io, procedural

PRE_UPLOAD = ("leaf_size", "max_bvh_depth", "node_pairs", "leaf_align", "bvh_builder", "ploc_radius", "wide_leaves")  # builder / layout options: before upload_scene
NOT_OPTIONS = ("spp", "census", "shard_rank", "shard_world", "shard_tile", "detail_u", "detail_v", "chain")


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "c4"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    opts = dict(a.split("=") for a in sys.argv[3:])
    ctx = B.Context(0)
    for k, v in opts.items():
        if k in PRE_UPLOAD:
            ctx.set_option(k, int(v))
    if which == "c4":
        _, mats = scene_io.parse_scene(os.path.join(ROOT, "assets", "dragon.json"))
        du = int(opts.get("detail_u", 4357)); dv = int(opts.get("detail_v", 100))
        ents = scene_io.build_entities(procedural.dragon_standin(du, dv), mats)
        W, H, spp = 1920, 1080, int(opts.get("spp", 1024))
        cam = B.to_camera_data([4, 2.5, 0], [0, .75, 0], [0, 1, 0], 50, W, H)
    elif which == "c3":
        _, mats = scene_io.parse_scene(os.path.join(ROOT, "assets", "mitsuba.json"))
        ents = scene_io.build_entities(procedural.mitsuba_standin(), mats)
        W, H, spp = 1024, 1024, int(opts.get("spp", 512))
        cam = B.to_camera_data([4, 2.5, 0], [0, 0.75, 0], [0, 1, 0], 50, W, H)
    elif which == "c5":
        _, mats = scene_io.parse_scene(os.path.join(ROOT, "assets", "car.json"))
        ents = scene_io.build_entities(procedural.car_standin(), mats)
        W, H, spp = 1920, 1080, int(opts.get("spp", 4096))
        cam = B.to_camera_data([0, 2, 5], [0, 0.5, 0], [0, 1, 0], 45, W, H)
    else:
        sc = scene_io.load_scene_dir(os.path.join(ROOT, "assets"), "cornell-box")
        mats, ents = sc["materials"], sc["entities"]
        W, H, spp = 512, 512, int(opts.get("spp", 256))
        c = sc["camera"]
        cam = B.to_camera_data(c["look_from"], c["look_at"], c["look_up"], c["vertical_fov"], W, H)
    if which == "c5":
        gi = [n for n, _, _ in mats].index("Ground")
        envmap = procedural.rgbe_to_ldr_rgba8(procedural.synthetic_sky_rgbe(2048, 1024))
        ctx.upload_scene(ents, [m for _, m, _ in mats], textures=[scene_io.checker_texture(256, 256, 16)], mesh_textures=[0 if mid == gi else -1 for _, mid in ents],
                         env=B.make_env(use_map=True, intensity=1.0, env_map=envmap))
    elif which == "c3":
        ctx.upload_scene(ents, [m for _, m, _ in mats], env=B.make_env(use_auto=True, intensity=1.0))
    else:
        ctx.upload_scene(ents, [m for _, m, _ in mats], env=B.make_env(color=(1, 1, 1), intensity=0.0))
    if "shard_rank" in opts:
        ctx.set_pixel_shard(int(opts["shard_rank"]), int(opts.get("shard_world", 8)), int(opts.get("shard_tile", 16)))
    for k, v in opts.items():
        if k not in PRE_UPLOAD and k not in NOT_OPTIONS:
            ctx.set_option(k, int(v))
    ms = []
    for _ in range(reps):
        ctx.render(cam, W, H, spp, 16)
        ms.append(ctx.stats()["kernel_ms"])
    st = ctx.stats()
    if opts.get("count"):
        pass
    if "shard_rank" in opts or opts.get("census"):
        ctx.set_option("count", 1)
        ctx.render(cam, W, H, spp, 16)
        cs0 = ctx.stats()
        ctx.set_option("count", 0)
        print(json.dumps({"rays": cs0["rays"], "nodes": cs0["nodes"], "tris": cs0["tris"], "samples": cs0["samples"],
                          "ns_per_ray": round(min(ms) * 1e6 / max(1, cs0["rays"]), 4), "ns_per_node": round(min(ms) * 1e6 / max(1, cs0["nodes"]), 5)}))
    if opts.get("census"):
        ctx.set_option("count", 1)
        ctx.render(cam, W, H, spp, 16)
        cs = ctx.stats()
        ctx.set_option("count", 0)
        sc = cs["sched"]
        names = ["node_steps", "node_lanes", "tri_steps", "tri_lanes", "retire_passes", "retired", "hit_passes", "hit_items", "miss_passes", "miss_items", "winddown_iters", "winddown_idle", "iters", "idle_sum", "donewait_sum", "active_sum", "wait_polls", "rays_after_death", "sleeps", "pushes", "pushes_ge8", "pushes_ge12", "pushes_ge16", "winddown_ticks"]
        cen = dict(zip(names, sc))
        cen.update({k: cs[k] for k in ("rays", "nodes", "tris", "scatters", "samples")})
        for a, b in (("node_lanes", "node_steps"), ("tri_lanes", "tri_steps"), ("retired", "retire_passes"), ("hit_items", "hit_passes"), ("miss_items", "miss_passes"), ("winddown_idle", "winddown_iters"), ("idle_sum", "iters"), ("donewait_sum", "iters"), ("active_sum", "iters")):
            cen[a + "/" + b] = round(cen[a] / max(1, cen[b]), 2)
        cyc = dict(zip(["node", "tri", "retire", "hit_pass", "miss_pass", "park_resume", "sleep", "total"], sc[24:32]))
        cen["winddown_ray_share"] = round(cen["rays_after_death"] / max(1, cen["rays"]), 4)
        cen["winddown_time_share"] = round(cen["winddown_ticks"] / max(1, cyc["total"]), 4)
        cen["cycle_share"] = {k: round(v / max(1, cyc["total"]), 4) for k, v in cyc.items()}
        g = cs["groups"]
        cen["groups"] = {"phases": g[0], "iters": g[1], "iters/phase": round(g[1] / max(1, g[0]), 2), "busy_groups/iter": round(g[2] / max(1, g[1]), 2),
                         "node_groups/iter": round(g[3] / max(1, g[1]), 2), "leaf_groups/iter": round(g[4] / max(1, g[1]), 2), "rays": g[5],
                         "ray_share": round(g[5] / max(1, cs["rays"]), 4), "iters/ray": round(g[2] / max(1, g[5]), 2),
                         "cycle_share": round(g[6] / max(1, cyc["total"]), 4), "cycles/iter": round(g[6] / max(1, g[1]), 1)}
        print(json.dumps(cen))
    if opts.get("chain"):
        # per-ray turnaround of the longest sample chain: the cost pre-pass counted the rays of the first 8 samples of every pixel
        # (saturating at 255), so the most expensive pixel traces about max_cost / 8 * spp sequential rays in kernel_ms
        _, _, cost = ctx.read_queue(W * H)
        if cost.size:
            mx = int(cost.max()); p999 = float(np.percentile(cost, 99.9))
            print(json.dumps({"chain": {"max_cost_per_8spp": mx, "p99.9_cost": p999, "longest_chain_rays": int(mx / 8 * spp),
                                        "us_per_ray_longest_chain": round(min(ms) * 1e3 / max(1.0, mx / 8 * spp), 2),
                                        "us_per_ray_p99.9_chain": round(min(ms) * 1e3 / max(1.0, p999 / 8 * spp), 2)}}))
    if opts.get("timeline"):
        print(json.dumps({"laps_ms": ctx.read_laps()}))
    print(json.dumps({"lib": os.path.basename(B.LIB_PATH), "scene": which, "opts": opts, "kernel_ms_min": round(min(ms), 2), "kernel_ms_med": round(float(np.median(ms)), 2),
                      "Msamples/s": round(W * H * spp / min(ms) / 1e3, 1), "vgprs": st["vgprs"], "lds": st["lds_bytes"], "grid": st["grid"], "block": st["block"],
                      "bvh_depth": st["bvh_depth"], "bvh_nodes": st["bvh_nodes"], "bvh_build_ms": round(st["bvh_build_ms"], 2)}))

main()
