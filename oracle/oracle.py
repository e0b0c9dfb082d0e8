"""ctypes wrapper around oracle/libpt_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product (owl-path-tracer_amd) never does.  See oracle/pt_oracle.h for the parity note
(PARITY UNPINNED: the reference ships no fixtures for this path and cannot be built here).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PT_ORACLE_LIB: the tolerance-calibration build (libpt_oracle_hostlibm.so, oracle/Makefile) in a process of its own
_LIB_PATH = os.environ.get("PT_ORACLE_LIB") or os.path.join(_HERE, "libpt_oracle.so")

MAT_FLOATS = 17
MAT_KEYS = [
    "base_color_r", "base_color_g", "base_color_b", "subsurface", "metallic", "specular", "specular_tint",
    "roughness", "anisotropic", "sheen", "sheen_tint", "clearcoat", "clearcoat_gloss", "ior",
    "specular_transmission", "specular_transmission_roughness", "emission",
]
# material_data{} defaults, device_global.hpp:21-35
MAT_DEFAULT = np.array([0.8, 0.8, 0.8, 0.0, 0.0, 0.5, 1.0, 0.5, 0.0, 0.0, 1.0, 0.0, 0.03, 1.45, 0.0, 0.0, 0.0], np.float32)

LOBE_NONE, LOBE_DIFFUSE, LOBE_CLEARCOAT, LOBE_METALLIC, LOBE_GLASS = -1, 0, 1, 2, 3

DM_FN = {"sin": 0, "cos": 1, "tan": 2, "atan": 3, "atan2": 4, "asin": 5, "log": 6, "exp": 7, "pow": 8, "sqrt": 9, "div": 10}


class Texture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgba8", C.POINTER(C.c_uint32))]


class SceneDesc(C.Structure):
    _fields_ = [
        ("n_tris", C.c_int32),
        ("positions", C.POINTER(C.c_float)),
        ("normals", C.POINTER(C.c_float)),
        ("texcoords", C.POINTER(C.c_float)),
        ("material_index", C.POINTER(C.c_int32)),
        ("texture_index", C.POINTER(C.c_int32)),
        ("n_materials", C.c_int32),
        ("materials", C.POINTER(C.c_float)),
        ("n_textures", C.c_int32),
        ("textures", C.POINTER(Texture)),
    ]


class Env(C.Structure):
    _fields_ = [("use_map", C.c_int32), ("use_auto", C.c_int32), ("color", C.c_float * 3), ("intensity", C.c_float), ("map", Texture)]


class Camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("llc", C.c_float * 3), ("horizontal", C.c_float * 3), ("vertical", C.c_float * 3)]

    def as_array(self):
        return np.array(list(self.origin) + list(self.llc) + list(self.horizontal) + list(self.vertical), np.float32)


class Counters(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("rays", "nodes", "tris", "scatters", "env_misses", "samples", "nan_retries")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


def build(force=False):
    """(Re)build the shared object with oracle/Makefile."""
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "pt_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    fp = C.POINTER(C.c_float)
    L.orc_scene_create.restype = C.c_void_p
    L.orc_scene_create.argtypes = [C.POINTER(SceneDesc), C.c_int]
    L.orc_scene_destroy.argtypes = [C.c_void_p]
    L.orc_scene_set_materials.argtypes = [C.c_void_p, fp, C.c_int]
    L.orc_scene_bvh_nodes.argtypes = [C.c_void_p]
    L.orc_scene_bvh_depth.argtypes = [C.c_void_p]
    L.orc_to_camera_data.argtypes = [fp, fp, fp, C.c_float, C.c_int, C.c_int, C.POINTER(Camera)]
    L.orc_intersect.argtypes = [C.c_void_p, fp, fp, C.c_float, C.c_float, C.c_int, fp, fp, fp, C.POINTER(C.c_int32)]
    L.orc_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Env), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                             C.POINTER(C.c_uint32), C.c_int64, fp, C.POINTER(C.c_uint32), C.POINTER(Counters)]
    L.orc_trace_pixel.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Env), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_int, fp, C.POINTER(C.c_uint32)]
    L.orc_rng_init.restype = C.c_uint32
    L.orc_rng_init.argtypes = [C.c_uint32, C.c_uint32]
    L.orc_rng_next.restype = C.c_float
    L.orc_rng_next.argtypes = [C.POINTER(C.c_uint32)]
    L.orc_make_rgba.restype = C.c_uint32
    L.orc_make_rgba.argtypes = [fp]
    L.orc_sample_disney.argtypes = [fp, fp, C.POINTER(C.c_uint32), C.POINTER(C.c_int32), fp, fp, fp]
    L.orc_onb.argtypes = [fp, fp, fp]
    L.orc_to_local.argtypes = [fp, fp, fp, fp, fp]
    L.orc_to_world.argtypes = [fp, fp, fp, fp, fp]
    L.orc_sample_cosine_hemisphere.argtypes = [C.c_float, C.c_float, fp]
    L.orc_refract.argtypes = [fp, fp, C.c_float, fp]
    L.orc_fresnel_equation.restype = C.c_float
    L.orc_fresnel_equation.argtypes = [fp, fp, C.c_float, C.c_float]
    L.orc_d_gtr1.restype = C.c_float
    L.orc_d_gtr1.argtypes = [fp, C.c_float]
    L.orc_d_gtr2.restype = C.c_float
    L.orc_d_gtr2.argtypes = [fp, C.c_float, C.c_float]
    L.orc_lambda.restype = C.c_float
    L.orc_lambda.argtypes = [fp, C.c_float, C.c_float]
    L.orc_eval_lobe.argtypes = [C.c_int, fp, fp, fp, fp, fp, fp]
    L.orc_eval_sheen.argtypes = [fp, fp, fp, fp]
    L.orc_uv_on_sphere.argtypes = [fp, fp]
    L.orc_tex_nearest.argtypes = [C.POINTER(Texture), C.c_float, C.c_float, fp]
    for n in ("sin", "cos", "tan", "atan", "asin", "log", "exp"):
        f = getattr(L, "orc_dm_" + n)
        f.restype = C.c_float
        f.argtypes = [C.c_float]
    for n in ("atan2", "pow"):
        f = getattr(L, "orc_dm_" + n)
        f.restype = C.c_float
        f.argtypes = [C.c_float, C.c_float]
    L.orc_dm_batch.argtypes = [C.c_int, fp, fp, fp, C.c_int64]
    _lib = L
    return L


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _vec3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def _tex(arr):
    """arr: (H, W) uint32 RGBA8 (row 0 = v=0) or None."""
    t = Texture()
    if arr is None:
        t.width = 0
        t.height = 0
        t.rgba8 = None
        return t, None
    a = np.ascontiguousarray(arr, dtype=np.uint32)
    t.width = a.shape[1]
    t.height = a.shape[0]
    t.rgba8 = a.ctypes.data_as(C.POINTER(C.c_uint32))
    return t, a


def dm(fn, x, y=None):
    """Evaluate a deterministic-libm function elementwise (float32)."""
    xa, xp = _f(np.atleast_1d(x))
    out = np.empty_like(xa)
    if y is not None:
        ya, yp = _f(np.broadcast_to(np.atleast_1d(y), xa.shape))
    else:
        ya, yp = None, None
    lib().orc_dm_batch(DM_FN[fn], xp, yp, out.ctypes.data_as(C.POINTER(C.c_float)), xa.size)
    return out


def rng_init(u, v):
    return int(lib().orc_rng_init(u & 0xFFFFFFFF, v & 0xFFFFFFFF))


def rng_next(state):
    s = C.c_uint32(state)
    f = lib().orc_rng_next(C.byref(s))
    return float(f), int(s.value)


def make_rgba(c):
    return int(lib().orc_make_rgba(_vec3(c)))


def to_camera_data(look_from, look_at, look_up, vfov, w, h):
    cam = Camera()
    lib().orc_to_camera_data(_vec3(look_from), _vec3(look_at), _vec3(look_up), float(vfov), int(w), int(h), C.byref(cam))
    return cam


def camera_from_array(a):
    a = np.asarray(a, np.float32).reshape(12)
    cam = Camera()
    for i in range(3):
        cam.origin[i], cam.llc[i], cam.horizontal[i], cam.vertical[i] = a[i], a[3 + i], a[6 + i], a[9 + i]
    return cam


def sample_disney(mat, wo, rng_state, lobe=LOBE_NONE):
    m, mp = _f(mat)
    f = (C.c_float * 3)()
    wi = (C.c_float * 3)()
    pdf = C.c_float(0)
    st = C.c_uint32(rng_state)
    lb = C.c_int32(lobe)
    lib().orc_sample_disney(mp, _vec3(wo), C.byref(st), C.byref(lb), f, wi, C.byref(pdf))
    return dict(f=np.array(f[:], np.float32), wi=np.array(wi[:], np.float32), pdf=np.float32(pdf.value), lobe=int(lb.value),
                state=int(st.value))


def eval_lobe(lobe, mat, wo, wh, wi):
    m, mp = _f(mat)
    f = (C.c_float * 3)()
    pdf = C.c_float(0)
    lib().orc_eval_lobe(lobe, mp, _vec3(wo), _vec3(wh), _vec3(wi), f, C.byref(pdf))
    return np.array(f[:], np.float32), np.float32(pdf.value)


def eval_sheen(mat, wo, wi):
    m, mp = _f(mat)
    f = (C.c_float * 3)()
    lib().orc_eval_sheen(mp, _vec3(wo), _vec3(wi), f)
    return np.array(f[:], np.float32)


def onb(n):
    t = (C.c_float * 3)()
    b = (C.c_float * 3)()
    lib().orc_onb(_vec3(n), t, b)
    return np.array(t[:], np.float32), np.array(b[:], np.float32)


def to_local(t, b, n, w):
    o = (C.c_float * 3)()
    lib().orc_to_local(_vec3(t), _vec3(b), _vec3(n), _vec3(w), o)
    return np.array(o[:], np.float32)


def to_world(t, b, n, w):
    o = (C.c_float * 3)()
    lib().orc_to_world(_vec3(t), _vec3(b), _vec3(n), _vec3(w), o)
    return np.array(o[:], np.float32)


def sample_cosine_hemisphere(u0, u1):
    o = (C.c_float * 3)()
    lib().orc_sample_cosine_hemisphere(u0, u1, o)
    return np.array(o[:], np.float32)


def refract(w, n, eta):
    o = (C.c_float * 3)()
    ok = lib().orc_refract(_vec3(w), _vec3(n), eta, o)
    return bool(ok), np.array(o[:], np.float32)


def fresnel_equation(i, m, eta_i, eta_t):
    return float(lib().orc_fresnel_equation(_vec3(i), _vec3(m), eta_i, eta_t))


def d_gtr1(wh, alpha):
    return float(lib().orc_d_gtr1(_vec3(wh), alpha))


def d_gtr2(wm, ax, ay):
    return float(lib().orc_d_gtr2(_vec3(wm), ax, ay))


def lambda_(w, ax, ay):
    return float(lib().orc_lambda(_vec3(w), ax, ay))


def uv_on_sphere(n):
    o = (C.c_float * 2)()
    lib().orc_uv_on_sphere(_vec3(n), o)
    return np.array(o[:], np.float32)


def tex_nearest(tex, u, v):
    t, keep = _tex(tex)
    o = (C.c_float * 3)()
    lib().orc_tex_nearest(C.byref(t), u, v, o)
    return np.array(o[:], np.float32)


def make_env(use_map=False, use_auto=False, color=(0, 0, 0), intensity=0.0, env_map=None):
    e = Env()
    e.use_map = int(bool(use_map))
    e.use_auto = int(bool(use_auto))
    for i in range(3):
        e.color[i] = float(color[i])
    e.intensity = float(intensity)
    t, keep = _tex(env_map)
    e.map = t
    e._keep = keep
    return e


class Scene:
    """Oracle scene built from a flattened triangle soup (see pyhost.scene_io.flatten_scene)."""

    def __init__(self, flat, leaf_size=4):
        self._keep = []
        d = SceneDesc()
        n = int(flat["positions"].shape[0])
        d.n_tris = n
        pos, d.positions = _f(flat["positions"].reshape(-1))
        nrm, d.normals = _f(flat["normals"].reshape(-1))
        self._keep += [pos, nrm]
        if flat.get("texcoords") is not None:
            tc, d.texcoords = _f(flat["texcoords"].reshape(-1))
            self._keep.append(tc)
        else:
            d.texcoords = None
        mi = np.ascontiguousarray(flat["material_index"], np.int32)
        ti = np.ascontiguousarray(flat["texture_index"], np.int32)
        d.material_index = mi.ctypes.data_as(C.POINTER(C.c_int32))
        d.texture_index = ti.ctypes.data_as(C.POINTER(C.c_int32))
        mats, d.materials = _f(np.asarray(flat["materials"], np.float32).reshape(-1))
        d.n_materials = mats.size // MAT_FLOATS
        texs = flat.get("textures") or []
        d.n_textures = len(texs)
        arr = (Texture * max(1, len(texs)))()
        for i, t in enumerate(texs):
            tt, keep = _tex(t)
            arr[i] = tt
            self._keep.append(keep)
        d.textures = arr
        self._keep += [mi, ti, mats, arr]
        self.n_tris = n
        self.h = lib().orc_scene_create(C.byref(d), leaf_size)

    def __del__(self):
        try:
            if self.h:
                lib().orc_scene_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def set_materials(self, mats):
        m, mp = _f(np.asarray(mats, np.float32).reshape(-1))
        lib().orc_scene_set_materials(self.h, mp, m.size // MAT_FLOATS)

    @property
    def bvh_nodes(self):
        return lib().orc_scene_bvh_nodes(self.h)

    @property
    def bvh_depth(self):
        return lib().orc_scene_bvh_depth(self.h)

    def intersect(self, org, direction, tmin=1e-3, tmax=1e10, use_bvh=True):
        t = C.c_float()
        u = C.c_float()
        v = C.c_float()
        p = C.c_int32()
        ok = lib().orc_intersect(self.h, _vec3(org), _vec3(direction), tmin, tmax, int(use_bvh), C.byref(t), C.byref(u), C.byref(v), C.byref(p))
        return bool(ok), float(t.value), float(u.value), float(v.value), int(p.value)

    def render(self, cam, env, W, H, spp, max_depth, use_bvh=True, threads=None, pixel_list=None, want_rgba8=False, want_counters=False,
               out=None):
        """Returns (rgb float32 (H, W, 3) in framebuffer order [row 0 = top], rgba8 or None, counters dict or None)."""
        if threads is None:
            threads = os.cpu_count() or 1
        rgb = out if out is not None else np.zeros((H, W, 3), np.float32)
        rgba = np.zeros((H, W), np.uint32) if want_rgba8 else None
        cnt = Counters() if want_counters else None
        pl = None
        npx = 0
        if pixel_list is not None:
            pl = np.ascontiguousarray(pixel_list, np.uint32)
            npx = pl.size
        rc = lib().orc_render(self.h, C.byref(cam), C.byref(env), W, H, spp, max_depth, int(use_bvh), int(threads),
                              pl.ctypes.data_as(C.POINTER(C.c_uint32)) if pl is not None else None, npx,
                              rgb.ctypes.data_as(C.POINTER(C.c_float)),
                              rgba.ctypes.data_as(C.POINTER(C.c_uint32)) if rgba is not None else None,
                              C.byref(cnt) if cnt is not None else None)
        if rc != 0:
            raise RuntimeError("orc_render failed: %d" % rc)
        return rgb, rgba, (cnt.as_dict() if cnt is not None else None)

    def trace_sample(self, cam, env, W, H, px, py, sample, max_depth, use_bvh=True, max_rows=256):
        """Per-bounce log of one sample of one pixel: (rows, 32) float32, layout in pt_oracle.c (ORC_LOG_FLOATS)."""
        log = np.zeros((max_rows, 32), np.float32)
        f = lib().orc_trace_sample
        f.restype = C.c_int
        n = f(C.c_void_p(self.h), C.byref(cam), C.byref(env), C.c_int(W), C.c_int(H), C.c_int(px), C.c_int(py), C.c_int(sample), C.c_int(max_depth), C.c_int(int(use_bvh)),
              log.ctypes.data_as(C.POINTER(C.c_float)), C.c_int(max_rows))
        return log[:n]

    def trace_pixel(self, cam, env, W, H, px, py, spp, max_depth, use_bvh=True):
        rgb = np.zeros((spp, 3), np.float32)
        st = np.zeros(spp, np.uint32)
        lib().orc_trace_pixel(self.h, C.byref(cam), C.byref(env), W, H, px, py, spp, max_depth, int(use_bvh),
                              rgb.ctypes.data_as(C.POINTER(C.c_float)), st.ctypes.data_as(C.POINTER(C.c_uint32)))
        return rgb, st
