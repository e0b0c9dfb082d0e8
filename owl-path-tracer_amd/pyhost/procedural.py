"""Deterministic procedural stand-ins for the reference's missing geometry blobs.

The reference's `assets/{dragon,mitsuba,car,sphere}.obj.scene` are absent (`.MISSING_LARGE_BLOBS`), so the
BASELINE configs that name them run on closed-form stand-ins that carry the SAME object names as the
materials in the reference's `assets/<scene>.json` (application.cpp:166-179 matches meshes to materials by
name).  Everything here is closed-form numpy (no RNG); results are float32.
Numbers measured on stand-ins are not comparable with renders of the original assets.
"""
import numpy as np


def _grid_indices(nu, nv, wrap_u=True, wrap_v=True):
    """Two triangles per (u, v) cell of an nu x nv vertex grid (row-major: id = iu*nv + iv)."""
    cu = nu if wrap_u else nu - 1
    cv = nv if wrap_v else nv - 1
    iu, iv = np.meshgrid(np.arange(cu), np.arange(cv), indexing="ij")
    iu1 = (iu + 1) % nu
    iv1 = (iv + 1) % nv
    a = iu * nv + iv
    b = iu1 * nv + iv
    c = iu1 * nv + iv1
    d = iu * nv + iv1
    t0 = np.stack([a, b, c], -1).reshape(-1, 3)
    t1 = np.stack([a, c, d], -1).reshape(-1, 3)
    return np.concatenate([t0, t1], 1).reshape(-1, 3).astype(np.int32)


def _mesh(vertices, normals, indices, texcoords=None):
    v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
    n = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
    tc = np.zeros((0, 2), np.float32) if texcoords is None else np.ascontiguousarray(texcoords, np.float32).reshape(-1, 2)
    return dict(vertices=v, normals=n, texcoords=tc, indices=np.ascontiguousarray(indices, np.int32).reshape(-1, 3))


def quad(p0, p1, p2, p3, normal, uv=False):
    v = np.array([p0, p1, p2, p3], np.float32)
    n = np.tile(np.asarray(normal, np.float32), (4, 1))
    idx = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    tc = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32) if uv else None
    return _mesh(v, n, idx, tc)


def uv_sphere(center, radius, nu=64, nv=32, flip=False):
    """Latitude/longitude sphere with duplicated pole rings (all faces are proper triangles)."""
    u = (np.arange(nu) / nu) * 2 * np.pi
    v = (np.arange(nv + 1) / nv) * np.pi
    uu, vv = np.meshgrid(u, v, indexing="ij")
    n = np.stack([np.sin(vv) * np.cos(uu), np.cos(vv), np.sin(vv) * np.sin(uu)], -1)
    p = np.asarray(center, np.float64) + radius * n
    idx = _grid_indices(nu, nv + 1, wrap_u=True, wrap_v=False)
    # drop degenerate triangles at the poles
    P = p.reshape(-1, 3)
    e1 = P[idx[:, 1]] - P[idx[:, 0]]
    e2 = P[idx[:, 2]] - P[idx[:, 0]]
    area = np.linalg.norm(np.cross(e1, e2), axis=1)
    idx = idx[area > 1e-12]
    nn = -n if flip else n
    return _mesh(P, nn.reshape(-1, 3), idx)


def knot_tube(n_u, n_v, center=(0.0, 0.85, 0.0), scale=0.75, p=2, q=3, tube=0.17):
    """Bumpy tube around a (p,q) torus knot: the high-poly 'dragon' stand-in body (2*n_u*n_v triangles)."""
    t = (np.arange(n_u, dtype=np.float64) / n_u) * 2 * np.pi
    s = (np.arange(n_v, dtype=np.float64) / n_v) * 2 * np.pi

    def curve(tt):
        r = 1.0 + 0.4 * np.cos(q * tt)
        return np.stack([r * np.cos(p * tt), 0.75 * np.sin(q * tt), r * np.sin(p * tt)], -1) * scale

    h = 1e-4
    C = curve(t)
    T = curve(t + h) - curve(t - h)
    T /= np.linalg.norm(T, axis=1, keepdims=True)
    n1 = np.stack([np.cos(q * t) * np.cos(p * t), np.sin(q * t), np.cos(q * t) * np.sin(p * t)], -1)
    N = n1 - (n1 * T).sum(1, keepdims=True) * T
    N /= np.linalg.norm(N, axis=1, keepdims=True)
    B = np.cross(T, N)
    tt, ss = np.meshgrid(t, s, indexing="ij")
    rho = tube * (1.0 + 0.16 * np.sin(17 * tt + 3 * ss) * np.sin(5 * ss) + 0.07 * np.sin(61 * tt) * np.cos(9 * ss)
                  + 0.035 * np.sin(233 * tt + 11 * ss))
    P = C[:, None, :] + rho[..., None] * (np.cos(ss)[..., None] * N[:, None, :] + np.sin(ss)[..., None] * B[:, None, :])
    P += np.asarray(center, np.float64)
    # smooth normals from central differences on the closed grid
    du = np.roll(P, -1, 0) - np.roll(P, 1, 0)
    dv = np.roll(P, -1, 1) - np.roll(P, 1, 1)
    nrm = np.cross(du, dv)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=-1, keepdims=True), 1e-30)
    # orient outward (away from the centre curve)
    outward = ((P - np.asarray(center, np.float64) - C[:, None, :]) * nrm).sum(-1, keepdims=True)
    nrm = np.where(outward < 0, -nrm, nrm)
    return _mesh(P.reshape(-1, 3), nrm.reshape(-1, 3), _grid_indices(n_u, n_v))


def dragon_standin(n_u=4357, n_v=100):
    """C4 stand-in: objects named as in assets/dragon.json -- 'dragon' (2*n_u*n_v = 871 400 tris by default,
    the size of the usual Stanford dragon), 'ground' quad, 'areaLight' quad (emission 30 in the JSON)."""
    dragon = knot_tube(n_u, n_v)
    ground = quad((-8, 0, -8), (-8, 0, 8), (8, 0, 8), (8, 0, -8), (0, 1, 0))
    light = quad((-1.0, 3.6, -1.0), (1.0, 3.6, -1.0), (1.0, 3.6, 1.0), (-1.0, 3.6, 1.0), (0, -1, 0))
    return [("dragon", dragon), ("ground", ground), ("areaLight", light)]


def mitsuba_standin(detail=96):
    """C3 stand-in (objects 'outside', 'inside', 'ground' as in assets/mitsuba.json): a sphere shell,
    an inner sphere and a ground disk-like quad.  ~4*detail^2 triangles."""
    outside = uv_sphere((0, 0.95, 0), 0.9, nu=2 * detail, nv=detail)
    inside = uv_sphere((0, 0.95, 0), 0.45, nu=2 * detail, nv=detail)
    # cut a window in the outer shell so the inner sphere is visible from the camera (+x side)
    P = outside["vertices"]
    idx = outside["indices"]
    cen = P[idx].mean(1)
    keep = ~((cen[:, 0] > 0.55) & (np.abs(cen[:, 1] - 0.95) < 0.42) & (np.abs(cen[:, 2]) < 0.42))
    outside = dict(outside, indices=np.ascontiguousarray(idx[keep]))
    ground = quad((-8, 0, -8), (-8, 0, 8), (8, 0, 8), (8, 0, -8), (0, 1, 0))
    return [("outside", outside), ("inside", inside), ("ground", ground)]


def furnace_sphere(detail=48):
    """Unit-test scene: the object 'sphere' of assets/sphere.json (geometry blob missing upstream)."""
    return [("sphere", uv_sphere((0, 1, 0), 1.0, nu=2 * detail, nv=detail))]


def write_obj(path, meshes):
    """Emit `.obj.scene` text (one 'o' per mesh, v/vt/vn or v//vn corners) so the normal loader path can be exercised."""
    with open(path, "w") as f:
        f.write("# procedural stand-in (owl-path-tracer_amd/pyhost/procedural.py)\n")
        vo, to, no = 1, 1, 1
        for name, m in meshes:
            f.write("o %s\n" % name)
            V, N, TC, I = m["vertices"], m["normals"], m["texcoords"], m["indices"]
            for v in V:
                f.write("v %.9g %.9g %.9g\n" % (v[0], v[1], v[2]))
            for t in TC:
                f.write("vt %.9g %.9g\n" % (t[0], t[1]))
            for n in N:
                f.write("vn %.9g %.9g %.9g\n" % (n[0], n[1], n[2]))
            has_tc = TC.shape[0] == V.shape[0] and V.shape[0] > 0
            for tri in I:
                if has_tc:
                    f.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % tuple(x for k in tri for x in (k + vo, k + to, k + no)))
                else:
                    f.write("f %d//%d %d//%d %d//%d\n" % tuple(x for k in tri for x in (k + vo, k + no)))
            vo += V.shape[0]
            to += TC.shape[0]
            no += N.shape[0]
