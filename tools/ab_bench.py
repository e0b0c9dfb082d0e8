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
NOT_OPTIONS = ("finish", "tiers", "spp", "census", "shard_rank", "shard_world", "shard_tile", "detail_u", "detail_v", "chain", "sweep", "frame_out")

# C3 material sweep ("for BSDF coverage add a material sweep over metallic/clearcoat/transmission/sheen"; the reference's
# driver is test_loop / modify_sbt, application.hpp:89-108, application.cpp:329-360): the attribute is set on the two objects of the
# mitsuba stand-in ('outside', 'inside'; 'ground' stays diffuse) through pt_set_materials, one frame per value - the per-lobe cost harness.
SWEEP = [("base", {}), ("metallic=1,roughness=0.3", {4: 1.0, 7: 0.3}), ("metallic=0.5", {4: 0.5}), ("clearcoat=1", {11: 1.0}),
         ("specular_transmission=1", {14: 1.0}), ("specular_transmission=0.5,tr_roughness=0.3", {14: 0.5, 15: 0.3}), ("sheen=1", {9: 1.0}),
         ("all four lobes: metallic .3, clearcoat 1, transmission .5, sheen .5", {4: 0.3, 11: 1.0, 14: 0.5, 9: 0.5})]
LOBE_NAMES = ["diffuse", "clearcoat", "metallic", "glass", "emitter", "nan_retry"]


def lobe_census(cs):
    lb = cs["lobes"]
    tot = max(1, sum(lb[:6]))
    out = {"items": {n: lb[i] for i, n in enumerate(LOBE_NAMES)}, "share": {n: round(lb[i] / tot, 4) for i, n in enumerate(LOBE_NAMES)},
           "lanes_per_body_execution": {n: round(lb[i] / max(1, lb[8 + i]), 2) for i, n in enumerate(LOBE_NAMES) if lb[i]},
           "hit_passes": cs["sched"][6], "bodies_per_pass": round(sum(lb[8:12]) / max(1, cs["sched"][6]), 3),
           "passes_with_2+_bodies": round(lb[14] / max(1, cs["sched"][6]), 4), "passes_single_branch": round(lb[15] / max(1, cs["sched"][6]), 4)}
    return out


def cached_meshes(name, make):
    """The procedural stand-ins take up to 18 s to generate in numpy: keep them in /tmp for the later invocations of the same GPU-box session."""
    path = "/tmp/ptamd_mesh_%s.npz" % name
    if os.path.exists(path):
        z = np.load(path, allow_pickle=False)
        names = [str(x) for x in z["names"]]
        return [(n, dict(vertices=z["v%d" % i], normals=z["n%d" % i], texcoords=z["t%d" % i], indices=z["i%d" % i])) for i, n in enumerate(names)]
    ms = make()
    d = {"names": np.array([n for n, _ in ms])}
    for i, (_, m) in enumerate(ms):
        d["v%d" % i] = m["vertices"]; d["n%d" % i] = m["normals"]; d["i%d" % i] = m["indices"]
        d["t%d" % i] = m["texcoords"]
    try:
        np.savez(path + ".tmp.npz", **d)
        os.replace(path + ".tmp.npz", path)
    except OSError:
        pass
    return ms


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
        ents = scene_io.build_entities(cached_meshes("dragon_%d_%d" % (du, dv), lambda: procedural.dragon_standin(du, dv)), mats)
        W, H, spp = 1920, 1080, int(opts.get("spp", 1024))
        cam = B.to_camera_data([4, 2.5, 0], [0, .75, 0], [0, 1, 0], 50, W, H)
    elif which == "c3":
        _, mats = scene_io.parse_scene(os.path.join(ROOT, "assets", "mitsuba.json"))
        ents = scene_io.build_entities(procedural.mitsuba_standin(), mats)
        W, H, spp = 1024, 1024, int(opts.get("spp", 512))
        cam = B.to_camera_data([4, 2.5, 0], [0, 0.75, 0], [0, 1, 0], 50, W, H)
    elif which == "c5":
        _, mats = scene_io.parse_scene(os.path.join(ROOT, "assets", "car.json"))
        ents = scene_io.build_entities(cached_meshes("car", procedural.car_standin), mats)
        W, H, spp = 1920, 1080, int(opts.get("spp", 4096))
        cam = B.to_camera_data([0, 2, 5], [0, 0.5, 0], [0, 1, 0], 45, W, H)
    else:
        sc = scene_io.load_scene_dir(os.path.join(ROOT, "assets"), "cornell-box")
        mats, ents = sc["materials"], sc["entities"]
        W, H, spp = 512, 512, int(opts.get("spp", 256))
        c = sc["camera"]
        cam = B.to_camera_data(c["look_from"], c["look_at"], c["look_up"], c["vertical_fov"], W, H)
    t_up = time.perf_counter()
    if which == "c5":
        gi = [n for n, _, _ in mats].index("Ground")
        envmap = procedural.rgbe_to_ldr_rgba8(procedural.synthetic_sky_rgbe(2048, 1024))
        ctx.upload_scene(ents, [m for _, m, _ in mats], textures=[scene_io.checker_texture(256, 256, 16)], mesh_textures=[0 if mid == gi else -1 for _, mid in ents],
                         env=B.make_env(use_map=True, intensity=1.0, env_map=envmap))
    elif which == "c3":
        ctx.upload_scene(ents, [m for _, m, _ in mats], env=B.make_env(use_auto=True, intensity=1.0))
    else:
        ctx.upload_scene(ents, [m for _, m, _ in mats], env=B.make_env(color=(1, 1, 1), intensity=0.0))
    upload_ms = (time.perf_counter() - t_up) * 1e3
    if "shard_rank" in opts:
        ctx.set_pixel_shard(int(opts["shard_rank"]), int(opts.get("shard_world", 8)), int(opts.get("shard_tile", 16)))
    for k, v in opts.items():
        if k not in PRE_UPLOAD and k not in NOT_OPTIONS:
            ctx.set_option(k, int(v))
    if opts.get("sweep"):  # c3 sweep=1: one timed frame (+ one counted frame) per material variant
        base = np.stack([m for _, m, _ in mats]).astype(np.float32)
        names = [n for n, _, _ in mats]
        for label, edits in SWEEP:
            mm = base.copy()
            for i, n in enumerate(names):
                if n != "ground":
                    for k, v in edits.items():
                        mm[i, k] = v
            ctx.set_materials(mm)
            t = []
            for _ in range(reps):
                ctx.render(cam, W, H, spp, 16)
                t.append(ctx.stats()["kernel_ms"])
            ctx.set_option("count", 1)
            ctx.render(cam, W, H, spp, 16)
            cs = ctx.stats()
            ctx.set_option("count", 0)
            cyc = dict(zip(["node", "tri", "retire", "hit_pass", "miss_pass", "park_resume", "sleep", "total"], cs["sched"][24:32]))
            print(json.dumps({"sweep": label, "kernel_ms_min": round(min(t), 2), "Msamples/s": round(W * H * spp / min(t) / 1e3, 1), "rays_per_sample": round(cs["rays"] / max(1, cs["samples"]), 3),
                              "nodes_per_ray": round(cs["nodes"] / max(1, cs["rays"]), 2), "tris_per_ray": round(cs["tris"] / max(1, cs["rays"]), 2), "nan_retries": cs["nan_retries"],
                              "cycle_share": {k: round(v / max(1, cyc["total"]), 4) for k, v in cyc.items()}, "hit_items/pass": round(cs["sched"][7] / max(1, cs["sched"][6]), 2),
                              "lobes": lobe_census(cs)}))
        return
    ms = []
    for _ in range(reps):
        rgb, _ = ctx.render(cam, W, H, spp, 16)
        ms.append(ctx.stats()["kernel_ms"])
    st = ctx.stats()
    if opts.get("frame_out"):  # crc of the float frame: A/B runs must agree bit for bit
        import zlib
        print(json.dumps({"frame_crc32": "%08x" % (zlib.crc32(np.ascontiguousarray(rgb, np.float32).tobytes()) & 0xFFFFFFFF)}))
    if opts.get("tiers"):
        print(json.dumps({"tiers(pixels,per_wave,waves,class)": [(t["pixels"], t["per_wave"], t["waves"], t["cost_class"]) for t in ctx.read_tiers()]}))
    if opts.get("finish"):
        # when was each pixel done (needs latency=1)?  How many pixels are still running as the frame drains, and how fast are the last ones served?
        q, ids, cost = ctx.read_queue(W * H)
        fin, nrays = ctx.read_finish(W * H)
        if fin.size and ids.size:
            f = fin[ids.astype(np.int64)]
            total = float(f.max())
            rays = np.maximum(cost.astype(np.float64), 1.0) / 8.0 * (spp - 8)
            if nrays.any():  # count=1: the real thing
                est = rays
                rays = np.maximum(nrays[ids.astype(np.int64)].astype(np.float64), 1.0)
                print(json.dumps({"true_rays_over_estimate(p1,p10,p50,p90,p99)": [round(float(x), 2) for x in np.percentile(rays / est, [1, 10, 50, 90, 99])],
                                  "pixels_with_rays_ge": {str(k): int((rays >= k).sum()) for k in (1500, 2000, 2500, 3000, 4000, 5000, 6000, 8000)}}))
            us = f * 1e3 / rays
            nexp = ctx.stats().get("express_pixels", 0)
            pos = np.empty(W * H, np.int64); pos[:] = 1 << 40
            pos[q.astype(np.int64)] = np.arange(q.size)
            is_exp = pos[ids.astype(np.int64)] < nexp
            order = np.argsort(-f)
            out = {"pixels_with_cost_ge": {str(k): int((cost >= k).sum()) for k in (8, 12, 16, 20, 24, 32, 40, 48, 64)}, "last_pixel_ms": round(total, 2), "n_pixels": int(ids.size), "n_express": int(nexp),
                   "running_at_pct_of_time": {str(pc): int((f > total * pc / 100.0).sum()) for pc in (10, 20, 30, 40, 50, 60, 70, 80, 90, 95, 99)},
                   "rays_left_at_pct_of_time": {str(pc): round(float((rays * np.clip((f - total * pc / 100.0) / np.maximum(f, 1e-9), 0, 1)).sum() / rays.sum()), 4) for pc in (10, 20, 30, 40, 50, 60, 70, 80, 90)},
                   "last_finishers(cost,ms,us_per_ray,express)": [(int(cost[i]), round(float(f[i]), 1), round(float(us[i]), 1), bool(is_exp[i])) for i in order[:12]],
                   "top_cost(cost,ms,us_per_ray,express)": [(int(cost[i]), round(float(f[i]), 1), round(float(us[i]), 1), bool(is_exp[i])) for i in np.argsort(-cost.astype(np.int64))[:12]],
                   "us_per_ray_express(p10,p50,p90)": [round(float(x), 1) for x in np.percentile(us[is_exp], [10, 50, 90])] if is_exp.any() else None,
                   "us_per_ray_bulk_by_cost_class(cost>>5: p50)": {str(k): round(float(np.median(us[(~is_exp) & ((cost >> 5) == k)])), 1) for k in range(8) if ((~is_exp) & ((cost >> 5) == k)).any()},
                   "finish_ms_express(p10,p50,p90,max)": [round(float(x), 1) for x in np.percentile(f[is_exp], [10, 50, 90, 100])] if is_exp.any() else None}
            print(json.dumps(out))
            tiers = ctx.read_tiers()
            if tiers:  # per tier: pixels per wave, true rays (median), finish time (median, max), us per ray (median)
                fq = fin[q.astype(np.int64)]; rq = np.maximum(nrays[q.astype(np.int64)].astype(np.float64), 1.0) if nrays.any() else None
                rows = []
                for t in tiers:
                    sl = slice(t["q0"], t["q0"] + t["pixels"])
                    if t["pixels"] < 20 or rq is None: continue
                    ff = fq[sl]; rr = rq[sl]; late = ff > 1.5 * np.median(ff)
                    rows.append((t["cost_class"], t["pixels"], t["per_wave"], int(np.median(rr)), round(float(np.median(ff)), 1), round(float(np.percentile(ff, 90)), 1), round(float(np.percentile(ff, 99)), 1), round(float(ff.max()), 1),
                                 round(float(np.median(ff * 1e3 / rr)), 1), int(late.sum()), int(np.median(rr[late])) if late.any() else 0, round(float(np.median(ff[late] * 1e3 / rr[late])), 1) if late.any() else 0))
                print(json.dumps({"per_tier(class,pixels,per_wave,rays_p50,finish_p50,p90,p99,max,us_per_ray_p50,n_late(>1.5x p50),late_rays_p50,late_us_per_ray_p50)": rows}))
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
        cen["nodes_per_ray"] = round(cs["nodes"] / max(1, cs["rays"]), 2)
        cen["tris_per_ray"] = round(cs["tris"] / max(1, cs["rays"]), 2)
        cen["rays_per_sample"] = round(cs["rays"] / max(1, cs["samples"]), 3)
        cen["env_misses"] = cs["env_misses"]
        cen["lobes"] = lobe_census(cs)
        tv = cs["trav"]
        quad_lane_steps = max(1, cen["node_lanes"])  # lanes in node rounds x ~steps per round: use nodes / 2 (a quad step counts two node units)
        cen["trav"] = {"quad_lane_steps": cs["nodes"] // 2, "no_child_entered": tv[0], "of_those_beyond_best_hit": tv[1], "leaf_lane_steps": cen["tri_lanes"], "leaf_steps_without_improvement": tv[3],
                       "share_no_child": round(tv[0] / max(1, cs["nodes"] // 2), 4), "share_cullable": round(tv[1] / max(1, cs["nodes"] // 2), 4),
                       "share_leaf_no_improvement": round(tv[3] / max(1, cen["tri_lanes"]), 4)}
        g = cs["groups"]
        cen["groups"] = {"phases": g[0], "iters": g[1], "iters/phase": round(g[1] / max(1, g[0]), 2), "busy_groups/iter": round(g[2] / max(1, g[1]), 2),
                         "node_groups/iter": round(g[3] / max(1, g[1]), 2), "leaf_groups/iter": round(g[4] / max(1, g[1]), 2), "rays": g[5],
                         "ray_share": round(g[5] / max(1, cs["rays"]), 4), "iters/ray": round(g[2] / max(1, g[5]), 2),
                         "cycle_share": round(g[6] / max(1, cyc["total"]), 4), "cycles/iter": round(g[6] / max(1, g[1]), 1)}
        print(json.dumps(cen))
    if opts.get("chain"):
        # per-ray turnaround of the longest sample chain: one more render that counts the rays of every pixel (latency=1)
        ctx.set_option("latency", 1)
        ctx.render(cam, W, H, spp, 16)
        fin, nrays = ctx.read_finish(W * H)
        ctx.set_option("latency", 0)
        if nrays.size and nrays.any():
            mx = int(nrays.max()); p999 = float(np.percentile(nrays[nrays > 0], 99.9)); last = int(np.argmax(fin))
            print(json.dumps({"chain": {"longest_chain_rays": mx, "p99.9_chain_rays": int(p999), "us_per_ray_longest_chain": round(min(ms) * 1e3 / max(1, mx), 2),
                                        "us_per_ray_p99.9_chain": round(min(ms) * 1e3 / max(1.0, p999), 2),
                                        "last_pixel(rays,ms)": (int(nrays[last]), round(float(fin[last]), 1)), "tiers": len(ctx.read_tiers()), "prepass_spp": ctx.stats().get("prepass_spp", 0)}}))
    if opts.get("timeline"):
        print(json.dumps({"laps_ms": ctx.read_laps()}))
    print(json.dumps({"lib": os.path.basename(B.LIB_PATH), "scene": which, "opts": opts, "kernel_ms_min": round(min(ms), 2), "kernel_ms_med": round(float(np.median(ms)), 2),
                      "Msamples/s": round(W * H * spp / min(ms) / 1e3, 1), "prepass_ms": round(st.get("prepass_ms", 0.0), 2), "vgprs": st["vgprs"], "lds": st["lds_bytes"], "grid": st["grid"], "block": st["block"],
                      "bvh_depth": st["bvh_depth"], "bvh_nodes": st["bvh_nodes"], "bvh_build_ms": round(st["bvh_build_ms"], 2), "upload_scene_ms": round(upload_ms, 1)}))

if __name__ == "__main__":
    main()
