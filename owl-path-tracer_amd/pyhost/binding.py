"""ctypes binding of the C-ABI in include/mi355pt.h (libmi355pt.so) for tests and bench.py.

There is NO fallback: if the shared object is missing, or a render is requested without a gfx950 GPU, this
raises.  Nothing here imports or calls the CPU oracle.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("PT_LIB_PATH") or os.path.join(_PKG, "libmi355pt.so")  # PT_LIB_PATH: A/B builds (tools/ab_bench.py)
HEADER_PATH = os.path.join(os.path.dirname(_PKG), "include", "mi355pt.h")

PT_MAT_FLOATS = 17
OPS = {"sin": 0, "cos": 1, "tan": 2, "atan": 3, "atan2": 4, "asin": 5, "log": 6, "exp": 7, "pow": 8, "sqrt": 9, "div": 10,
       "sample_disney": 20, "closest_hit": 21, "frame": 22, "rng": 23}


class PtError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("reserved", C.c_int32)]


class Mesh(C.Structure):
    _fields_ = [("vertices", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)), ("texcoords", C.POINTER(C.c_float)),
                ("indices", C.POINTER(C.c_int32)), ("n_vertices", C.c_int32), ("n_normals", C.c_int32), ("n_texcoords", C.c_int32),
                ("n_triangles", C.c_int32), ("material_index", C.c_int32), ("texture_index", C.c_int32)]


class Texture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgba8", C.POINTER(C.c_uint32))]


class Env(C.Structure):
    _fields_ = [("use_map", C.c_int32), ("use_auto", C.c_int32), ("color", C.c_float * 3), ("intensity", C.c_float), ("map", Texture)]


class Camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("llc", C.c_float * 3), ("horizontal", C.c_float * 3), ("vertical", C.c_float * 3)]

    def as_array(self):
        return np.array(list(self.origin) + list(self.llc) + list(self.horizontal) + list(self.vertical), np.float32)


class Stats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("launches", C.c_int32), ("vgprs", C.c_int32), ("sgprs", C.c_int32), ("lds_bytes", C.c_int32),
                ("block", C.c_int32), ("grid", C.c_int32), ("stack_entries", C.c_int32),
                ("samples", C.c_uint64), ("rays", C.c_uint64), ("nodes", C.c_uint64), ("tris", C.c_uint64), ("scatters", C.c_uint64),
                ("env_misses", C.c_uint64), ("nan_retries", C.c_uint64), ("bvh_nodes", C.c_uint64), ("bvh_depth", C.c_uint64),
                ("n_triangles", C.c_uint64), ("bvh_build_ms", C.c_double), ("sched", C.c_uint64 * 32), ("prepass_ms", C.c_double), ("groups", C.c_uint64 * 8), ("reduce_ms", C.c_double), ("d2h_ms", C.c_double), ("kernel_variant", C.c_int32), ("express_pixels", C.c_int32), ("whole_pixels", C.c_int32), ("prepass_spp", C.c_int32), ("lobes", C.c_uint64 * 16), ("trav", C.c_uint64 * 4)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["sched"] = list(self.sched)
        d["groups"] = list(self.groups)
        d["lobes"] = list(self.lobes)
        d["trav"] = list(self.trav)
        return d


EXPORTS = ["pt_create", "pt_destroy", "pt_last_error", "pt_abi_version", "pt_upload_scene", "pt_set_materials", "pt_set_environment",
           "pt_set_pixel_shard", "pt_shard_pixels", "pt_render", "pt_render_device", "pt_synchronize", "pt_set_option", "pt_get_stats",
           "pt_to_camera_data", "pt_debug_closest_hit_host", "pt_debug_eval", "pt_debug_read_queue", "pt_debug_read_laps", "pt_debug_read_finish", "pt_debug_read_tiers", "pt_debug_plan_tiers",
           "pt_comm_get_unique_id", "pt_comm_init_rank", "pt_comm_destroy", "pt_reduce_framebuffer", "pt_host_alloc", "pt_host_free",
           "pt_group_create", "pt_group_destroy", "pt_group_size", "pt_group_ctx", "pt_group_last_error", "pt_group_upload_scene",
           "pt_group_set_materials", "pt_group_set_option", "pt_group_render", "pt_debug_quad_info", "pt_debug_oct_info", "pt_debug_clone_scene"]
PT_COMM_ID_BYTES = 128

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PtError("libmi355pt.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` (no CPU fallback exists)")
    L = C.CDLL(LIB_PATH)
    fp = C.POINTER(C.c_float)
    L.pt_create.restype = C.c_void_p
    L.pt_create.argtypes = [C.POINTER(Config)]
    L.pt_destroy.restype = None
    L.pt_destroy.argtypes = [C.c_void_p]
    L.pt_last_error.restype = C.c_char_p
    L.pt_last_error.argtypes = [C.c_void_p]
    L.pt_abi_version.restype = C.c_int
    L.pt_upload_scene.argtypes = [C.c_void_p, C.POINTER(Mesh), C.c_int32, fp, C.c_int32, C.POINTER(Texture), C.c_int32,
                                  C.POINTER(C.c_int32), C.POINTER(Env)]
    L.pt_set_materials.argtypes = [C.c_void_p, fp, C.c_int32]
    L.pt_set_environment.argtypes = [C.c_void_p, C.POINTER(Env)]
    L.pt_set_pixel_shard.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    L.pt_shard_pixels.restype = C.c_int64
    L.pt_shard_pixels.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint32), C.c_int64]
    L.pt_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_int32, C.c_int32, C.c_int32, C.c_int32, fp, C.POINTER(C.c_uint32)]
    L.pt_render_device.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pt_synchronize.argtypes = [C.c_void_p]
    L.pt_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.pt_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
    L.pt_to_camera_data.restype = None
    L.pt_to_camera_data.argtypes = [fp, fp, fp, C.c_float, C.c_int32, C.c_int32, C.POINTER(Camera)]
    L.pt_debug_closest_hit_host.argtypes = [C.c_void_p, fp, fp, C.c_float, C.c_float, fp, fp, fp, C.POINTER(C.c_int32)]
    L.pt_debug_eval.argtypes = [C.c_void_p, C.c_int32, fp, C.c_int32, fp, C.c_int32, C.c_int64]
    L.pt_debug_read_queue.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8), C.c_int64]
    L.pt_debug_read_queue.restype = C.c_int64
    L.pt_debug_read_laps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int64]
    L.pt_debug_read_laps.restype = C.c_int64
    L.pt_debug_read_finish.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_int64]
    L.pt_debug_read_finish.restype = C.c_int64
    L.pt_debug_read_tiers.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_int64]
    L.pt_debug_read_tiers.restype = C.c_int64
    L.pt_debug_plan_tiers.argtypes = [C.POINTER(C.c_uint32), C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint32), C.c_int64]
    L.pt_debug_plan_tiers.restype = C.c_int64
    u8p = C.POINTER(C.c_uint8)
    L.pt_comm_get_unique_id.argtypes = [u8p]
    L.pt_comm_init_rank.argtypes = [C.c_void_p, u8p, C.c_int32, C.c_int32]
    L.pt_comm_destroy.argtypes = [C.c_void_p]
    L.pt_reduce_framebuffer.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.pt_host_alloc.restype = C.c_void_p
    L.pt_host_alloc.argtypes = [C.c_size_t]
    L.pt_host_free.restype = None
    L.pt_host_free.argtypes = [C.c_void_p]
    L.pt_group_create.restype = C.c_void_p
    L.pt_group_create.argtypes = [C.POINTER(C.c_int32), C.c_int32]
    L.pt_group_destroy.restype = None
    L.pt_group_destroy.argtypes = [C.c_void_p]
    L.pt_group_size.argtypes = [C.c_void_p]
    L.pt_group_ctx.restype = C.c_void_p
    L.pt_group_ctx.argtypes = [C.c_void_p, C.c_int32]
    L.pt_group_last_error.restype = C.c_char_p
    L.pt_group_last_error.argtypes = [C.c_void_p]
    L.pt_group_upload_scene.argtypes = [C.c_void_p, C.POINTER(Mesh), C.c_int32, fp, C.c_int32, C.POINTER(Texture), C.c_int32,
                                        C.POINTER(C.c_int32), C.POINTER(Env)]
    L.pt_group_set_materials.argtypes = [C.c_void_p, fp, C.c_int32]
    L.pt_group_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.pt_debug_quad_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    L.pt_debug_oct_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    L.pt_debug_clone_scene.argtypes = [C.c_void_p, C.c_void_p]
    L.pt_group_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_int32, C.c_int32, C.c_int32, C.c_int32, fp, C.POINTER(C.c_uint32)]
    _lib = L
    return L


def comm_unique_id():
    """128 opaque bytes from rank 0 (ncclGetUniqueId inside the library) for pt_comm_init_rank on every rank."""
    buf = (C.c_uint8 * PT_COMM_ID_BYTES)()
    rc = lib().pt_comm_get_unique_id(buf)
    if rc < 0:
        raise PtError("pt_comm_get_unique_id failed (%d): %s" % (rc, lib().pt_last_error(None).decode()))
    return bytes(buf)


class PinnedFrame:
    """W x H x 3 float32 (and optionally W x H uint32) in pinned host memory from pt_host_alloc, viewed as numpy arrays."""

    def __init__(self, W, H, want_rgba8=False):
        self._p = lib().pt_host_alloc(W * H * 12)
        self._p8 = lib().pt_host_alloc(W * H * 4) if want_rgba8 else None
        if not self._p or (want_rgba8 and not self._p8):
            raise PtError("pt_host_alloc failed")
        self.rgb = np.ctypeslib.as_array(C.cast(self._p, C.POINTER(C.c_float)), shape=(H, W, 3))
        self.rgba8 = np.ctypeslib.as_array(C.cast(self._p8, C.POINTER(C.c_uint32)), shape=(H, W)) if want_rgba8 else None

    def free(self):
        if getattr(self, "_p", None):
            self.rgb = None
            lib().pt_host_free(self._p)
            self._p = None
        if getattr(self, "_p8", None):
            self.rgba8 = None
            lib().pt_host_free(self._p8)
            self._p8 = None


def _vec3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def to_camera_data(look_from, look_at, look_up, vfov, w, h):
    cam = Camera()
    lib().pt_to_camera_data(_vec3(look_from), _vec3(look_at), _vec3(look_up), float(vfov), int(w), int(h), C.byref(cam))
    return cam


def shard_pixels(w, h, tile, rank, world):
    n = lib().pt_shard_pixels(w, h, tile, rank, world, None, 0)
    if n < 0:
        raise PtError("pt_shard_pixels: invalid arguments")
    ids = np.empty(int(n), np.uint32)
    lib().pt_shard_pixels(w, h, tile, rank, world, ids.ctypes.data_as(C.POINTER(C.c_uint32)), n)
    return ids


def _texture(arr):
    t = Texture()
    if arr is None:
        t.width = t.height = 0
        t.rgba8 = None
        return t, None
    a = np.ascontiguousarray(arr, np.uint32)
    t.width, t.height = a.shape[1], a.shape[0]
    t.rgba8 = a.ctypes.data_as(C.POINTER(C.c_uint32))
    return t, a


def make_env(use_map=False, use_auto=False, color=(0, 0, 0), intensity=0.0, env_map=None):
    e = Env()
    e.use_map = int(bool(use_map))
    e.use_auto = int(bool(use_auto))
    for i in range(3):
        e.color[i] = float(color[i])
    e.intensity = float(intensity)
    e.map, e._keep = _texture(env_map)
    return e


def plan_tiers(bucket_pixels, capacity, ns=96, force=False):
    """The device's tier plan (pt_tiers.h) for a cost histogram of 32 buckets, run on the host: list of dicts as Context.read_tiers()."""
    h = np.ascontiguousarray(bucket_pixels, np.uint32)
    assert h.size == 32
    t = np.zeros(257, np.uint32)
    n = lib().pt_debug_plan_tiers(h.ctypes.data_as(C.POINTER(C.c_uint32)), int(capacity), int(ns), int(bool(force)), t.ctypes.data_as(C.POINTER(C.c_uint32)), t.size)
    if n < 0:
        raise RuntimeError("pt_debug_plan_tiers: %d" % n)
    return [dict(zip(("q0", "pixels", "per_wave", "wave0", "waves", "cost_class"), (int(x) for x in t[1 + 8 * i:7 + 8 * i]))) for i in range(int(t[0]))]


class Context:
    """Thin object wrapper; device=-1 gives a host-only validation context (no render possible)."""

    def __init__(self, device=0):
        cfg = Config(device, 0)
        self._h = lib().pt_create(C.byref(cfg))
        if not self._h:
            raise PtError("pt_create failed: " + lib().pt_last_error(None).decode())
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            lib().pt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            raise PtError("%s failed (%d): %s" % (what, rc, lib().pt_last_error(self._h).decode()))
        return rc

    def set_option(self, key, value):
        self._check(lib().pt_set_option(self._h, key.encode(), int(value)), "pt_set_option")

    def upload_scene(self, entities, materials, textures=None, mesh_textures=None, env=None):
        """entities: list of (mesh dict, material index); materials: (n,17) float32;
        textures: list of (H,W) uint32 arrays; mesh_textures: per-entity texture index (or None)."""
        textures = textures or []
        keep = []
        arr = (Mesh * max(1, len(entities)))()
        for i, (m, mat_id) in enumerate(entities):
            v = np.ascontiguousarray(m["vertices"], np.float32)
            n = np.ascontiguousarray(m["normals"], np.float32)
            tc = np.ascontiguousarray(m["texcoords"], np.float32)
            idx = np.ascontiguousarray(m["indices"], np.int32)
            keep += [v, n, tc, idx]
            e = arr[i]
            e.vertices = v.ctypes.data_as(C.POINTER(C.c_float))
            e.normals = n.ctypes.data_as(C.POINTER(C.c_float)) if n.size else None
            e.texcoords = tc.ctypes.data_as(C.POINTER(C.c_float)) if tc.size else None
            e.indices = idx.ctypes.data_as(C.POINTER(C.c_int32))
            e.n_vertices, e.n_normals, e.n_texcoords, e.n_triangles = v.shape[0], n.shape[0], tc.shape[0], idx.shape[0]
            e.material_index = int(mat_id)
            e.texture_index = int(mesh_textures[i]) if mesh_textures is not None else -1
        mats = np.ascontiguousarray(np.asarray(materials, np.float32).reshape(-1, PT_MAT_FLOATS))
        tarr = (Texture * max(1, len(textures)))()
        for i, t in enumerate(textures):
            tarr[i], k = _texture(t)
            keep.append(k)
        envp = C.byref(env) if env is not None else None
        self._check(lib().pt_upload_scene(self._h, arr, len(entities), mats.ctypes.data_as(C.POINTER(C.c_float)), mats.shape[0], tarr,
                                          len(textures), None, envp), "pt_upload_scene")

    def set_materials(self, materials):
        mats = np.ascontiguousarray(np.asarray(materials, np.float32).reshape(-1, PT_MAT_FLOATS))
        self._check(lib().pt_set_materials(self._h, mats.ctypes.data_as(C.POINTER(C.c_float)), mats.shape[0]), "pt_set_materials")

    def set_environment(self, env):
        self._check(lib().pt_set_environment(self._h, C.byref(env)), "pt_set_environment")

    def set_pixel_shard(self, rank, world, tile=16):
        self._check(lib().pt_set_pixel_shard(self._h, rank, world, tile), "pt_set_pixel_shard")

    def render(self, cam, W, H, spp, max_depth, want_rgba8=False):
        rgb = np.empty((H, W, 3), np.float32)
        rgba = np.empty((H, W), np.uint32) if want_rgba8 else None
        self._check(lib().pt_render(self._h, C.byref(cam), W, H, spp, max_depth, rgb.ctypes.data_as(C.POINTER(C.c_float)),
                                    rgba.ctypes.data_as(C.POINTER(C.c_uint32)) if rgba is not None else None), "pt_render")
        return rgb, rgba

    def render_into(self, cam, W, H, spp, max_depth, rgb, rgba8=None):
        """pt_render into caller-owned arrays (e.g. a PinnedFrame); rgb may be None on the non-root ranks of a communicator."""
        self._check(lib().pt_render(self._h, C.byref(cam), W, H, spp, max_depth, rgb.ctypes.data_as(C.POINTER(C.c_float)) if rgb is not None else None,
                                    rgba8.ctypes.data_as(C.POINTER(C.c_uint32)) if rgba8 is not None else None), "pt_render")

    def comm_init_rank(self, unique_id, rank, world):
        buf = (C.c_uint8 * PT_COMM_ID_BYTES).from_buffer_copy(unique_id)
        self._check(lib().pt_comm_init_rank(self._h, buf, rank, world), "pt_comm_init_rank")

    def comm_destroy(self):
        self._check(lib().pt_comm_destroy(self._h), "pt_comm_destroy")

    def reduce_framebuffer(self, d_rgb, d_rgba8, n_pixels, stream=None):
        self._check(lib().pt_reduce_framebuffer(self._h, C.c_void_p(d_rgb), C.c_void_p(d_rgba8) if d_rgba8 else None, n_pixels,
                                                C.c_void_p(stream) if stream else None), "pt_reduce_framebuffer")

    def render_device(self, cam, W, H, spp, max_depth, d_out_rgb, d_out_rgba8=None, stream=None):
        self._check(lib().pt_render_device(self._h, C.byref(cam), W, H, spp, max_depth, C.c_void_p(d_out_rgb),
                                           C.c_void_p(d_out_rgba8) if d_out_rgba8 else None, C.c_void_p(stream) if stream else None),
                    "pt_render_device")

    def synchronize(self):
        self._check(lib().pt_synchronize(self._h), "pt_synchronize")

    def stats(self):
        s = Stats()
        self._check(lib().pt_get_stats(self._h, C.byref(s)), "pt_get_stats")
        return s.as_dict()

    def closest_hit_host(self, org, direction, tmin=1e-3, tmax=1e10):
        t, u, v, p = C.c_float(), C.c_float(), C.c_float(), C.c_int32()
        rc = self._check(lib().pt_debug_closest_hit_host(self._h, _vec3(org), _vec3(direction), tmin, tmax, C.byref(t), C.byref(u), C.byref(v),
                                                         C.byref(p)), "pt_debug_closest_hit_host")
        return bool(rc), float(t.value), float(u.value), float(v.value), int(p.value)

    def clone_scene_from(self, other):
        self._check(lib().pt_debug_clone_scene(self._h, other._h), "pt_debug_clone_scene")

    def quad_info(self):
        a = (C.c_int64 * 8)()
        self._check(lib().pt_debug_quad_info(self._h, a), "pt_debug_quad_info")
        return dict(zip(("quad_nodes", "depth", "leaf_slots", "triangles", "empty_slots", "internal_slots", "binary_nodes", "binary_leaf_refs"), [int(x) for x in a]))

    def oct_info(self):
        a = (C.c_int64 * 8)()
        self._check(lib().pt_debug_oct_info(self._h, a), "pt_debug_oct_info")
        return dict(zip(("oct_nodes", "depth", "leaf_slots", "triangles", "empty_slots", "internal_slots", "largest_leaf", "triangle_slots"), [int(x) for x in a]))

    def read_queue(self, cap):
        """(queue_ids, input_ids, cost) of the last cost-ordered render (empty arrays if it did not sort)."""
        q = np.zeros(cap, np.uint32); i = np.zeros(cap, np.uint32); c = np.zeros(cap, np.uint8)
        n = lib().pt_debug_read_queue(self._h, q.ctypes.data_as(C.POINTER(C.c_uint32)), i.ctypes.data_as(C.POINTER(C.c_uint32)),
                                      c.ctypes.data_as(C.POINTER(C.c_uint8)), cap)
        if n < 0:
            self._check(int(n), "pt_debug_read_queue")
        return q[:n], i[:n], c[:n]

    def read_finish(self, n_pixels):
        """Per pixel (x + W * y): (ms from the entry of the main launch to the pixel's last sample, rays traced - with count = 1); option latency = 1."""
        t = np.zeros(2 * n_pixels, np.uint32)
        n = lib().pt_debug_read_finish(self._h, t.ctypes.data_as(C.POINTER(C.c_uint32)), t.size)
        if n < 0:
            self._check(int(n), "pt_debug_read_finish")
        if n < 2 * n_pixels:
            return t[:0].astype(np.float64), t[:0]
        return t[:n_pixels].astype(np.float64) / 1e5, t[n_pixels:]

    def read_tiers(self):
        """Tiers of the last whole-pixel launch: list of dicts (first queue entry, pixels, pixels per wave, first workgroup, workgroups, cost class)."""
        t = np.zeros(257, np.uint32)
        n = lib().pt_debug_read_tiers(self._h, t.ctypes.data_as(C.POINTER(C.c_uint32)), t.size)
        if n < 0:
            self._check(int(n), "pt_debug_read_tiers")
        if n == 0:
            return []
        return [dict(zip(("q0", "pixels", "per_wave", "wave0", "waves", "cost_class"), (int(x) for x in t[1 + 8 * i:7 + 8 * i]))) for i in range(int(t[0]))]

    def read_laps(self):
        """ms since kernel entry at which the last pixel finished chunk 0, 1, ... of the last wavefront launch."""
        t = np.zeros(3 * 257 + 128, np.uint64)
        n = lib().pt_debug_read_laps(self._h, t.ctypes.data_as(C.POINTER(C.c_uint64)), t.size)
        if n < 0:
            self._check(int(n), "pt_debug_read_laps")
        m = (int(n) - 64) // 3
        ms = lambda x: round((int(x) - int(t[0])) / 1e5, 2)
        return {"last_done_ms": [ms(x) for x in t[1:m]], "last_start_ms": [ms(x) for x in t[m + 1:2 * m]], "last_entry": [int(x) for x in t[2 * m + 1:3 * m]],
                "first_chunk_ticks_by_cost_class": [int(x) for x in t[3 * m:3 * m + 32]], "first_chunk_count_by_cost_class": [int(x) for x in t[3 * m + 32:3 * m + 64]]}

    def debug_eval(self, op, inputs, out_stride):
        x = np.ascontiguousarray(inputs, np.float32)
        if x.ndim == 1:
            x = x[:, None]
        out = np.empty((x.shape[0], out_stride), np.float32)
        self._check(lib().pt_debug_eval(self._h, OPS[op] if isinstance(op, str) else op, x.ctypes.data_as(C.POINTER(C.c_float)), x.shape[1],
                                        out.ctypes.data_as(C.POINTER(C.c_float)), out_stride, x.shape[0]), "pt_debug_eval")
        return out
