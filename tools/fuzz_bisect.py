"""Find the first sample of a differing pixel of a fuzz case where product and oracle part ways."""
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
c = F.draw_case(seed, len(sys.argv) > 2)
ents, mats, W, H, spp, depth, mode, env = c["ents"], c["mats"], c["W"], c["H"], c["spp"], c["depth"], c["mode"], c["env"]
texs, mesh_tex, tex_by_mat = c["texs"], c["mesh_tex"], c["tex_by_mat"]
frm, at, up, fov = c["camera"]
print("case", seed, W, H, spp, depth, "env mode", mode, {k: (v if k != "env_map" else v.shape) for k, v in env.items()}, "tex", None if texs is None else texs[0].shape)
print("camera", frm, at, up, fov)
print("materials\n", mats)
cam = B.to_camera_data(frm, at, up, fov, W, H)
ocam = orc.to_camera_data(tuple(frm), tuple(at), tuple(up), fov, W, H)
gpu = B.Context(0)
gpu.upload_scene(ents, mats, textures=texs, mesh_textures=mesh_tex, env=B.make_env(**env))
S = orc.Scene(scene_io.flatten_scene(ents, [("m%d" % i, m, "") for i, m in enumerate(mats)], tex_by_mat))
oenv = orc.make_env(**env)
got, _ = gpu.render(cam, W, H, spp, depth)
want, _, _ = S.render(ocam, oenv, W, H, spp, depth)
bad = np.argwhere((got.view(np.uint32) != want.view(np.uint32)).any(axis=2))
print("differing pixels (row, col):", bad.tolist())
for (y, x) in bad[:3]:
    py = H - 1 - y
    first = None
    for s in range(1, spp + 1):
        g, _ = gpu.render(cam, W, H, s, depth)
        w_, _, _ = S.render(ocam, oenv, W, H, s, depth)
        if (g[y, x].view(np.uint32) != w_[y, x].view(np.uint32)).any():
            first = s
            print("pixel x=%d y=%d (py=%d): first difference with %d samples: gpu %s oracle %s" % (x, y, py, s, g[y, x], w_[y, x]))
            gm, _ = gpu.render(cam, W, H, s - 1, depth) if s > 1 else (np.zeros_like(g), None)
            wm, _, _ = S.render(ocam, oenv, W, H, s - 1, depth) if s > 1 else (np.zeros_like(g), None, None)
            print("   sample %d alone (difference of sums): gpu %s oracle %s" % (s - 1, g[y, x].astype(np.float64) * s - gm[y, x].astype(np.float64) * (s - 1), w_[y, x].astype(np.float64) * s - wm[y, x].astype(np.float64) * (s - 1)))
            break
    rgb, st = S.trace_pixel(ocam, oenv, W, H, int(x), int(py), spp, depth)
    if first:
        print("   oracle per-sample rgb around it:", rgb[max(0, first - 2):first + 1].tolist(), "rng states", st[max(0, first - 2):first + 1].tolist())
    # depth dependence: at which max depth does the difference appear (with `first` samples)?
    for d in range(1, depth + 1):
        g, _ = gpu.render(cam, W, H, first, d)
        w_, _, _ = S.render(ocam, oenv, W, H, first, d)
        if (g[y, x].view(np.uint32) != w_[y, x].view(np.uint32)).any():
            print("   appears from max depth", d)
            break

# ---- per-bounce replay of the first differing sample: the oracle's log, each bounce's ray through the product's closest-hit op and
# each bounce's (material, local_wo, rng) through the product's sample_disney op
for (y, x) in bad[:1]:
    py = H - 1 - y
    log = S.trace_sample(ocam, oenv, W, H, int(x), int(py), first - 1, depth)
    print("oracle log of sample %d: %d bounces" % (first - 1, len(log)))
    rays = log[:, 0:6].copy()
    hits = gpu.debug_eval("closest_hit", rays, 5)
    rows = []
    for r in log:
        mi = int(r[11:12].view(np.int32)[0])
        m = mats[mi] if 0 <= mi < len(mats) else mats[0]
        rows.append(np.concatenate([m, r[13:16], r[12:13], np.array([-1], np.int32).view(np.float32)]))
    sd = gpu.debug_eval("sample_disney", np.stack(rows).astype(np.float32), 9)
    prev_lobe = -1
    for k, r in enumerate(log):
        got_hit = bool(hits[k, 0])
        prim = int(hits[k, 4:5].view(np.int32)[0])
        oprim = int(r[10:11].view(np.int32)[0])
        same_hit = got_hit == bool(r[6]) and (not got_hit or ((hits[k, 1:4].view(np.uint32) == r[7:10].view(np.uint32)).all() and prim == oprim))
        line = "bounce %2d depth %2d org %s dir %s | oracle hit %d t %.9g u %.9g v %.9g prim %d | product hit %d t %.9g u %.9g v %.9g prim %d %s" % (
            k, int(r[31]), r[0:3], r[3:6], int(r[6]), r[7], r[8], r[9], oprim, got_hit, hits[k, 1], hits[k, 2], hits[k, 3], prim, "" if same_hit else "<<< HIT DIFFERS")
        print(line)
        if r[6]:
            # the op's input lobe: the previous bounce's sampled lobe matters? (sample_disney takes the lobe as in/out) - pass the oracle's previous one
            rows[k][-1] = np.array([prev_lobe], np.int32).view(np.float32)[0]
            sdk = gpu.debug_eval("sample_disney", rows[k][None, :].astype(np.float32), 9)[0]
            want = np.concatenate([r[16:19], r[20:23], r[19:20]])
            ok = (sdk[:7].view(np.uint32) == want.view(np.uint32)).all() and int(sdk[7:8].view(np.int32)[0]) == int(r[23:24].view(np.int32)[0]) and int(sdk[8:9].view(np.uint32)[0]) == int(r[27:28].view(np.uint32)[0])
            print("          material %d rng %08x local_wo %s | oracle f %s pdf %.9g wi %s lobe %d rng' %08x | product f %s pdf %.9g wi %s lobe %d rng' %08x %s" % (
                int(r[11:12].view(np.int32)[0]), int(r[12:13].view(np.uint32)[0]), r[13:16], r[16:19], r[19], r[20:23], int(r[23:24].view(np.int32)[0]), int(r[27:28].view(np.uint32)[0]),
                sdk[0:3], sdk[6], sdk[3:6], int(sdk[7:8].view(np.int32)[0]), int(sdk[8:9].view(np.uint32)[0]), "" if ok else "<<< SAMPLE DIFFERS"))
            prev_lobe = int(r[23:24].view(np.int32)[0])
