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


def uv_sphere(center, radius, nu=64, nv=32, flip=False, scale=(1.0, 1.0, 1.0)):
    """Latitude/longitude sphere (or axis-aligned ellipsoid via `scale`) with duplicated pole rings (all faces proper triangles)."""
    u = (np.arange(nu) / nu) * 2 * np.pi
    v = (np.arange(nv + 1) / nv) * np.pi
    uu, vv = np.meshgrid(u, v, indexing="ij")
    n = np.stack([np.sin(vv) * np.cos(uu), np.cos(vv), np.sin(vv) * np.sin(uu)], -1)
    sc = np.asarray(scale, np.float64)
    p = np.asarray(center, np.float64) + radius * n * sc
    n = n / sc  # ellipsoid normal ~ (x/sx^2, y/sy^2, z/sz^2)
    n = n / np.linalg.norm(n, axis=-1, keepdims=True)
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


def torus(center, R, r, nu=128, nv=48, axis=0):
    """Torus of major radius R / minor radius r around coordinate axis `axis` (wheel stand-in)."""
    u = (np.arange(nu) / nu) * 2 * np.pi
    v = (np.arange(nv) / nv) * 2 * np.pi
    uu, vv = np.meshgrid(u, v, indexing="ij")
    a = (R + r * np.cos(vv)) * np.cos(uu)
    b = (R + r * np.cos(vv)) * np.sin(uu)
    c = r * np.sin(vv)
    na, nb, nc = np.cos(vv) * np.cos(uu), np.cos(vv) * np.sin(uu), np.sin(vv)
    order = {0: (2, 0, 1), 1: (0, 2, 1), 2: (0, 1, 2)}[axis]  # which of (a, b, c) goes to x, y, z
    comp, ncomp = (a, b, c), (na, nb, nc)
    P = np.stack([comp[order[0]], comp[order[1]], comp[order[2]]], -1) + np.asarray(center, np.float64)
    N = np.stack([ncomp[order[0]], ncomp[order[1]], ncomp[order[2]]], -1)
    return _mesh(P.reshape(-1, 3), N.reshape(-1, 3), _grid_indices(nu, nv))


def car_standin(detail=1.0):
    """C5 stand-in carrying all 12 material names of assets/car.json (car.obj.scene is a missing blob): body ellipsoid with
    clearcoat paint, glass canopy (rough-glass lobe, inside/outside transitions), anisotropic carbon, metals, tyres, interior
    parts seen through the glass, textured ground (uv), emissive light.  1.72 M triangles at detail=1."""
    d = lambda n: max(8, int(round(n * detail)))
    parts = [
        ("BodyMat", uv_sphere((0, 0.62, 0), 1.0, d(1200), d(600), scale=(0.95, 0.42, 2.1))),
        ("WindowGlassMat", uv_sphere((0, 1.02, -0.15), 1.0, d(256), d(128), scale=(0.7, 0.4, 1.05))),
        ("Interior_Red", uv_sphere((-0.3, 0.95, -0.1), 0.22, d(96), d(48))),
        ("Interior_Black", uv_sphere((0.3, 0.95, -0.1), 0.22, d(96), d(48))),
        ("BodyMat_BK", uv_sphere((0, 0.45, 2.05), 0.35, d(128), d(64), scale=(2.2, 0.5, 0.5))),
        ("BodyGlossBlackMat", uv_sphere((0, 0.45, -2.05), 0.35, d(128), d(64), scale=(2.2, 0.5, 0.5))),
        ("CarbonBlack", uv_sphere((0, 1.25, -1.75), 0.3, d(128), d(64), scale=(2.6, 0.15, 0.6))),
        ("Default", uv_sphere((1.6, 0.3, 1.2), 0.3, d(96), d(48))),
    ]
    tyres, hubs = [], []
    for sx in (-1, 1):
        for sz in (-1.25, 1.25):
            tyres.append(torus((sx * 0.88, 0.36, sz), 0.24, 0.12, d(256), d(64), axis=0))
            hubs.append(uv_sphere((sx * 0.9, 0.36, sz), 0.16, d(64), d(32), scale=(0.5, 1, 1)))
    parts.append(("TireMat", merge_meshes(tyres)))
    parts.append(("EngineSilver2", merge_meshes(hubs)))
    parts.append(("Ground", quad((-10, 0, -10), (-10, 0, 10), (10, 0, 10), (10, 0, -10), (0, 1, 0), uv=True)))
    parts.append(("Light", quad((-1.5, 5.0, -1.5), (1.5, 5.0, -1.5), (1.5, 5.0, 1.5), (-1.5, 5.0, 1.5), (0, -1, 0))))
    return parts


def merge_meshes(ms):
    v, n, tc, idx, off = [], [], [], [], 0
    for m in ms:
        v.append(m["vertices"]); n.append(m["normals"]); tc.append(m["texcoords"])
        idx.append(m["indices"] + off)
        off += m["vertices"].shape[0]
    return dict(vertices=np.concatenate(v), normals=np.concatenate(n), texcoords=np.concatenate(tc), indices=np.concatenate(idx).astype(np.int32))


def synthetic_sky_rgbe(width=2048, height=1024):
    """Closed-form lat-long sky + sun as Radiance RGBE bytes (H, W, 4) (stand-in for the missing assets/environment.hdr)."""
    v, u = np.mgrid[0:height, 0:width]
    el = (0.5 - (v + 0.5) / height) * np.pi          # +pi/2 at the top row
    az = ((u + 0.5) / width - 0.5) * 2 * np.pi
    d = np.stack([np.cos(el) * np.sin(az), np.sin(el), np.cos(el) * np.cos(az)], -1)
    sun = np.array([0.4, 0.7, 0.59]); sun /= np.linalg.norm(sun)
    mu = (d * sun).sum(-1)
    t = np.clip(d[..., 1], 0, 1)
    sky = (1 - t)[..., None] * np.array([0.9, 0.9, 0.95]) + t[..., None] * np.array([0.25, 0.45, 0.95])
    ground = np.array([0.18, 0.16, 0.14])
    rgb = np.where(d[..., 1:2] >= 0, sky, ground) + (np.exp((mu - 1) * 2000.0) * 60.0 + np.exp((mu - 1) * 40.0) * 0.6)[..., None]
    m = rgb.max(-1)
    e = np.where(m > 1e-32, np.floor(np.log2(np.maximum(m, 1e-38))) + 1, 0)
    scale = np.where(m > 1e-32, np.ldexp(1.0, (8 - e).astype(np.int32)), 0.0)
    out = np.zeros((height, width, 4), np.uint8)
    out[..., :3] = np.clip(rgb * scale[..., None], 0, 255).astype(np.uint8)
    out[..., 3] = np.where(m > 1e-32, e + 128, 0).astype(np.uint8)
    return out


def write_hdr(path, rgbe):
    """Flat (non-RLE) Radiance file."""
    h, w = rgbe.shape[:2]
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w))
        f.write(np.ascontiguousarray(rgbe).tobytes())


def rgbe_to_ldr_rgba8(rgbe, flip=True):
    """stb_image's HDR->LDR conversion (extern/stb/stb_image.h:1864-1890: pow(x, 1/2.2)*255+0.5, clamp) followed by the
    reference's vertical flip (image_buffer.cpp:50-55): the (H, W) uint32 environment texture the render path samples."""
    f1 = np.where(rgbe[..., 3:4] == 0, 0.0, np.ldexp(1.0, rgbe[..., 3:4].astype(np.int32) - 136)).astype(np.float32)
    lin = rgbe[..., :3].astype(np.float32) * f1
    z = np.clip(np.power(lin, np.float32(1.0 / 2.2)) * np.float32(255) + np.float32(0.5), 0, 255).astype(np.uint32)
    img = (z[..., 0] | (z[..., 1] << 8) | (z[..., 2] << 16) | np.uint32(0xFF000000)).astype(np.uint32)
    return np.ascontiguousarray(img[::-1]) if flip else img


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
