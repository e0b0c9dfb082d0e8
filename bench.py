#!/usr/bin/env python3
"""bench.py -- headline benchmark of the render hot path on MI355X.

Workload (BASELINE.json metric "Msamples/sec at 1920x1080x1024spp", config C4): dragon.json materials/camera on the
documented 871 400-triangle stand-in for the missing dragon.obj.scene, 1920x1080, 1024 spp, max depth 16, black
environment (intensity 0), light from the 'areaLight' quad.  One "step" = one complete frame.

  python bench.py [--gpus N --steps K --warmup W]           # N = 1
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU; every rank holds a scene replica and renders the pixel tiles it owns (16x16 tiles dealt
round-robin; pixels are independent, samples of one pixel are NOT -- the reference threads one RNG stream through all
samples of a pixel), then ONE RCCL reduce (sum) of the float3 framebuffer to rank 0 -- inside the library (pt_comm_init_rank +
pt_render, include/mi355pt.h).  torch.distributed is the CONTROL plane only and runs over gloo (CPU): the 128-byte communicator id,
the barrier, the max over ranks and the per-rank timing table - so the only RCCL runtime in the process is the one the library
loads (PT_RCCL_PATH overrides which).  Total work is fixed => "strong".

Timed span of a step: pt_render = kernels + reduce + D2H of the complete float3 frame into pinned host memory on
rank 0.  After the timed loop every rank checks that no watchdog fired and rank 0 checks the frame against the checksum of the
parity-tested image (tests/golden/c4_frame_crc.json; `--write-golden` regenerates it, tests/test_gpu_parity.py pins the same
render against the oracle).  Prints one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver of this pool supports dmabuf IPC only (RCCL across processes fails with "hipIpcGetMemHandle: invalid argument" otherwise);
# already exported on the boxes - set before anything touches HIP in case a launcher dropped it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402

W, H, SPP, DEPTH = 1920, 1080, 1024, 16
PREPASS_SPP = 8  # samples per pixel of the cost pre-pass launch unless pt_stats says otherwise (16 when a tier plan is prepared: shards)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def build_workload():
    import ptamd

    ptamd.load()
    from owl_path_tracer_amd.pyhost import procedural, scene_io

    _, mats = scene_io.parse_scene(os.path.join(ROOT, "assets", "dragon.json"))
    meshes = _cached("dragon_standin", procedural.dragon_standin)
    ents = scene_io.build_entities(meshes, mats)
    return scene_io, mats, ents


def _cached(name, make):
    """The stand-in takes ~17 s to generate in numpy; profiling runs start bench.py eight times on one box: keep the arrays in /tmp
    (deterministic generator; nothing is read from the repository or the network)."""
    path = "/tmp/ptamd_bench_%s_%d.npz" % (name, os.getuid())
    try:
        if os.path.exists(path):
            z = np.load(path, allow_pickle=False)
            return [(str(n), dict(vertices=z["v%d" % i], normals=z["n%d" % i], texcoords=z["t%d" % i], indices=z["i%d" % i])) for i, n in enumerate(z["names"])]
    except Exception:  # unreadable cache (another user's file, a stale format): generate
        pass
    ms = make()
    tmp = "%s.%d.tmp.npz" % (path, os.getpid())  # the N ranks of one node write side by side: a name of its own each, then an atomic rename
    try:
        d = {"names": np.array([n for n, _ in ms])}
        for i, (_, m) in enumerate(ms):
            d["v%d" % i], d["n%d" % i], d["t%d" % i], d["i%d" % i] = m["vertices"], m["normals"], m["texcoords"], m["indices"]
        np.savez(tmp, **d)
        os.replace(tmp, path)
    except Exception:
        try:
            os.remove(tmp)
        except OSError:
            pass
    return ms


def algorithmic_bytes(st, n_pixels_out, textured_scatters=0):
    # per node 64 B, per triangle test 36 B, per scatter 152 B (+28 B textured), 4 B per env-map miss,
    # 12 B per framebuffer pixel
    return st["nodes"] * 64 + st["tris"] * 36 + st["scatters"] * 152 + textured_scatters * 28 + st["env_misses"] * 4 + n_pixels_out * 12


TRAFFIC_JSONS = [os.path.join(ROOT, "profiles", "r%02d_traffic.json" % r) for r in range(9, 2, -1)]  # the newest committed PMC summary whose kernel fingerprint matches
GOLDEN_CRC = os.path.join(ROOT, "tests", "golden", "c4_frame_crc.json")


def kernel_fingerprint():
    """sha256 of the render kernel's sources and of the Makefile (its code-generation flags): PMC numbers are only quoted for the
    build they were measured on."""
    import hashlib

    h = hashlib.sha256()
    for f in ("pt_kernel.hip", "pt_trace.h", "pt_device.h", "pt_types.h", "Makefile"):
        with open(os.path.join(ROOT, "owl-path-tracer_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def measured_pmc(world):
    """PMC summary of the main launch from rocprofv3 passes of THIS command (FETCH_SIZE and WRITE_SIZE in separate passes;
    FETCH_SIZE x 1024 is the byte count for this kernel's 64-byte gathers: profiles/r01_fetch_calibration.md).
    PMC cannot be collected from inside the timed run, so the figures are read from the committed summary of the profiling run
    (tools/profile_round.sh + tools/publish_profile.py write it together with the fingerprint of the kernel sources); None when the
    kernel has changed since, when no file matches, or for N > 1."""
    if world != 1:
        return None
    fp = kernel_fingerprint()
    for path in TRAFFIC_JSONS:
        try:
            with open(path) as f:
                d = json.load(f)
            if d.get("kernel_fingerprint") == fp:
                d["file"] = os.path.relpath(path, ROOT)
                return d
        except (OSError, ValueError, KeyError):
            continue
    return None


def frame_crc(rgb):
    import zlib

    return zlib.crc32(np.ascontiguousarray(rgb, np.float32).tobytes()) & 0xFFFFFFFF


def effective_cpus():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU box gives 16 of 256)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(scene_io, mats, ents, cam_arr, target_s=15.0):
    """Oracle (kind 'port': our CPU restatement; the reference has no CPU path and cannot be built) on the host cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc

    S = orc.Scene(scene_io.flatten_scene(ents, mats))
    cam = orc.camera_from_array(cam_arr)
    env = orc.make_env(color=(1, 1, 1), intensity=0.0)
    cores = effective_cpus()
    t0 = time.perf_counter()
    S.render(cam, env, W, H, 1, DEPTH, threads=cores)
    t1 = time.perf_counter() - t0
    spp = int(max(1, min(64, round(target_s / max(t1, 1e-3)))))
    t0 = time.perf_counter()
    S.render(cam, env, W, H, spp, DEPTH, threads=cores)
    dt = time.perf_counter() - t0
    return {"value": round(W * H * spp / dt / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "same scene/camera at %dx%d, %d spp (of %d), depth %d, %.1f s wall, %d threads; Msamples/s does not depend on spp"
                      % (W, H, spp, SPP, DEPTH, dt, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp-per-launch", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--write-golden", action="store_true", help="store the checksum of this run's frame as the expected one (after the parity tests passed)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import ptamd

    ptamd.load()
    from owl_path_tracer_amd.pyhost import distributed as D

    rank, local_rank, world = D.env_rank()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`" % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the render path)")
    # PT_BENCH_DEVICE (rehearsals only): every rank on ONE device - the N > 1 code path of this script on a one-GPU box, together with
    # PT_RCCL_PATH = the stub collective of tests/stub/fake_rccl.cpp; such a line carries "rehearsal": true and is not a measurement
    rehearsal_dev = os.environ.get("PT_BENCH_DEVICE")
    if rehearsal_dev is not None:
        local_rank = int(rehearsal_dev)
    torch.cuda.set_device(local_rank)
    D.init(backend="gloo")  # control plane on the CPU: no second RCCL communicator next to the library's

    scene_io, mats, ents = build_workload()  # also loads the package
    from owl_path_tracer_amd.pyhost import binding as B

    ctx = B.Context(local_rank)
    ctx.upload_scene(ents, [m for _, m, _ in mats], env=B.make_env(color=(1, 1, 1), intensity=0.0))
    if world > 1:
        # the library owns the communicator: rank 0 draws the id (ncclGetUniqueId), the launcher's group carries the 128 bytes
        box = [B.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        ctx.comm_init_rank(box[0], rank, world)  # also sets this rank's pixel shard (16x16 tiles, round-robin)
    if args.spp_per_launch:
        ctx.set_option("spp_per_launch", args.spp_per_launch)
    cam = B.to_camera_data([4.0, 2.5, 0.0], [0.0, 0.75, 0.0], [0.0, 1.0, 0.0], 50.0, W, H)

    frame = B.PinnedFrame(W, H) if rank == 0 else None  # pinned host memory, like the reference's framebuffer (owl.hpp:108-111)

    def step():
        # pt_render: this rank's tiles -> (N > 1) ONE RCCL reduce of the float3 framebuffer onto rank 0 -> D2H on rank 0; blocking
        ctx.render_into(cam, W, H, SPP, DEPTH, frame.rgb if rank == 0 else None)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # instrumented passes (untimed): work counters for the algorithmic-bytes figure - of this build's walk (records as fetched: a
    # quad node = two 64-byte units, an oct node of the sparse-wave group walk = four) and of the plain binary walk of the same tree
    # (one 64-byte node per visit, leaves of <= 4 triangles: SURVEY 8(d)'s unit, independent of how this build packs its records)
    ctx.set_option("count", 1)
    step()
    cst = ctx.stats()
    for k, v in (("quad", 0), ("groups", 0)):
        ctx.set_option(k, v)
    step()
    cst_bin = ctx.stats()
    for k, v in (("quad", 1), ("groups", 1)):
        ctx.set_option(k, v)
    ctx.set_option("count", 0)
    own_pixels = int(B.shard_pixels(W, H, D.TILE, rank, world).size)

    for _ in range(args.warmup):
        step()
    fence()
    kernel_ms, prepass_ms, step_ms, reduce_ms, d2h_ms = [], [], [], [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        step()
        step_ms.append((time.perf_counter() - ts) * 1e3)
        fst = ctx.stats()  # HIP events recorded on the launch stream: first launch .. end, and end of pre-pass + sort; raises if a watchdog fired
        kernel_ms.append(fst["kernel_ms"])
        prepass_ms.append(fst["prepass_ms"])
        reduce_ms.append(fst["reduce_ms"])
        d2h_ms.append(fst["d2h_ms"])
    fence()
    elapsed = time.perf_counter() - t0
    ctx.synchronize()  # PT_E_HIP -> PtError if any wave's watchdog fired during the timed frames: an incomplete image is not a result
    med_ms = float(np.median(step_ms))
    # what each rank spent where (medians over the timed steps): explains a scaling curve without another run
    mine = {"rank": rank, "pixels": own_pixels, "kernel_ms": round(float(np.median(kernel_ms)), 3), "reduce_ms": round(float(np.median(reduce_ms)), 3),
            "d2h_ms": round(float(np.median(d2h_ms)), 3), "step_ms": round(med_ms, 3), "rays": int(cst["rays"])}
    per_rank = [mine]
    if world > 1:
        t = torch.tensor([elapsed, med_ms], dtype=torch.float64)  # CPU tensor: gloo
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, med_ms = float(t[0].item()), float(t[1].item())
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    st = ctx.stats()

    crc = frame_crc(frame.rgb) if rank == 0 else None
    if rank == 0:
        if args.write_golden:
            with open(GOLDEN_CRC, "w") as f:
                json.dump({"workload": "C4 stand-in 1920x1080x1024spp depth 16 (bench.py)", "crc32_float3_frame": crc, "provenance": "bench.py --write-golden (re-run the opt-in whole-frame oracle comparison to tie it to the oracle again)",
                           "mean": float(np.mean(frame.rgb, dtype=np.float64))}, f)
        frame_check = "no golden checksum"
        if os.path.exists(GOLDEN_CRC):
            want = json.load(open(GOLDEN_CRC))["crc32_float3_frame"]
            if want != crc:
                raise SystemExit("bench.py: the rendered frame (crc32 %08x) is not the parity-tested image (crc32 %08x)" % (crc, want))
            frame_check = "crc32 %08x == tests/golden/c4_frame_crc.json" % crc

    if rank == 0:
        launches = max(1, st["launches"])
        k_ms = float(np.median(kernel_ms))  # per frame on this rank
        p_ms = float(np.median(prepass_ms))  # of which: cost pre-pass launch (PREPASS_SPP samples per pixel) + queue sort
        alg_frame = algorithmic_bytes(cst, own_pixels)  # this rank's frame, both launches
        # The dominant kernel launch is the main one (launch 2 of 2 per frame): samples PREPASS_SPP.. of every pixel.  Counted work
        # is per frame; per-sample work does not depend on the sample index, so the main launch carries (SPP - PREPASS_SPP) / SPP.
        pre_spp = int(st.get("prepass_spp", 0)) or PREPASS_SPP
        main_share = (SPP - pre_spp) / SPP if launches > 1 else 1.0
        alg_bytes = alg_frame * main_share
        main_ms = k_ms - p_ms
        achieved = alg_bytes / (main_ms * 1e-3) / 1e9
        alg_bin = algorithmic_bytes(cst_bin, own_pixels) * main_share  # binary-visit equivalents of the same rays
        achieved_bin = alg_bin / (main_ms * 1e-3) / 1e9
        samples = W * H * SPP * args.steps
        pmc = measured_pmc(world)
        pmc_traffic = int(pmc["traffic_bytes_per_launch"]) if pmc else None
        pmc_binding = pmc.get("binding") if pmc else None
        out = {
            "metric": "Msamples/sec at 1920x1080x1024spp",
            "value": round(samples / elapsed / 1e6, 2),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "ms_per_step_median": round(med_ms, 3),
            "timed_span": "pt_render: kernels + RCCL reduce (N > 1) + D2H of the float3 frame into pinned host memory on rank 0",
            "frame_check": frame_check,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "C4 dragon.json on the 871400-triangle stand-in (dragon.obj.scene is a missing blob), 1920x1080, 1024 spp, "
                                   "max_path_depth 16, environment intensity 0, areaLight emission 30",
                       "tiles": "%dx%d round-robin over %d rank(s), one RCCL reduce of the float3 framebuffer inside the library (pt_comm_init_rank)" % (D.TILE, D.TILE, world),
                       "kernel": "wavefront-scheduled megakernel; per frame: cost pre-pass launch (%d spp) + queue sort + one persistent main launch" % (int(st.get("prepass_spp", 0)) or PREPASS_SPP), "triangles": int(st["n_triangles"]), "bvh_nodes": int(st["bvh_nodes"]),
                       "bvh_depth": int(st["bvh_depth"])},
            # `achieved` = algorithmic bytes in ITS unit - 64 B per BVH2 node visit (the rays of this frame counted as a plain
            # binary walk of the same tree by the instrumented instance), 36 B per triangle test, 152 B per scatter, 12 B per pixel - over
            # the duration of the main launch.  These bytes are mostly served by the caches: `traffic` / `hbm_measured_*` is what the PMC
            # counters saw beyond L2, and `binding` what the same profile says the kernel actually waits for.
            "roofline": {"bound": "hbm", "achieved": round(achieved_bin, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved_bin / HBM_PEAK_GBS, 4),
                         "traffic": pmc_traffic,
                         "hbm_measured_gbs": round(pmc_traffic / (main_ms * 1e-3) / 1e9, 1) if pmc_traffic else None,
                         "hbm_measured_frac": round(pmc_traffic / (main_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if pmc_traffic else None,
                         "binding": pmc_binding,
                         "accounting": "achieved = SURVEY 8(d) algorithmic bytes (64 B per BVH2 node visit of the same rays on the same tree, counted by the "
                                       "instrumented one-level walk; 36 B per triangle test; 152 B per scatter; 12 B per pixel) / duration of the main launch; "
                                       "*_as_fetched = bytes of the records this build requests (quad node 2 x 64 B, oct node 4 x 64 B per visit): not an HBM figure "
                                       "(can exceed the peak: the working set is cache resident); traffic = FETCH_SIZE + WRITE_SIZE of the main launch (rocprofv3 "
                                       "PMC, separate passes, %s); hbm_measured_frac = traffic / main-launch time / peak"
                                       % (pmc["file"] if pmc else "no committed PMC summary matches this kernel build"),
                         "achieved_as_fetched": round(achieved, 1), "frac_as_fetched": round(achieved / HBM_PEAK_GBS, 4),
                         "algorithmic_bytes_per_launch": int(alg_bin), "algorithmic_bytes_per_launch_as_fetched": int(alg_bytes),
                         "counts_per_frame_rank0_binary_walk": {k: int(cst_bin[k]) for k in ("rays", "nodes", "tris", "scatters")},
                         "kernel": "pt_render_wave_kernel<false>, main launch", "kernel_ms_per_launch": round(main_ms, 3), "launches_per_step": launches,
                         "prepass_and_sort_ms": round(p_ms, 3), "kernel_ms_per_frame": round(k_ms, 3),
                         "counts_per_frame_rank0": {k: int(cst[k]) for k in ("samples", "rays", "nodes", "tris", "scatters", "env_misses")},
                         "vgprs": st["vgprs"], "lds_bytes": st["lds_bytes"], "grid": st["grid"], "block": st["block"]},
        }
        out["per_rank"] = per_rank
        if rehearsal_dev is not None:
            out["rehearsal"] = True
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene_io, mats, ents, cam.as_array())
        print(json.dumps(out), flush=True)

    if frame is not None:
        frame.free()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
