"""Scene ingestion for the test/bench harness (Python glue above the C-ABI).

Mirrors the reference's host-side loaders so that the harness can feed both the HIP library and the
oracle with identical arrays:
  parse_settings   <- path_tracer/src/utils/parser.cpp:81-117
  parse_scene      <- parser.cpp:19-63 (materials), :65-78 (camera)
  load_obj         <- utils/mesh_loader.cpp:85-121 (+ tinyobjloader shape splitting)
  create_mesh      <- mesh_loader.cpp:9-83 (re-index by POSITION index only; first-seen normals/texcoords)
  build_entities   <- application.cpp:166-179 (mesh kept only if a material has the same name)
The C++ host (owl-path-tracer_amd/host) implements the same ingestion natively; tests cross-check both.
"""
import json
import os

import numpy as np

MAT_FLOATS = 17
_MAT_FIELDS = ["subsurface", "metallic", "specular", "specular_tint", "roughness", "anisotropic", "sheen", "sheen_tint", "clearcoat",
               "clearcoat_gloss", "ior", "specular_transmission", "specular_transmission_roughness", "emission"]
# material_data{} defaults, device_global.hpp:21-35
MAT_DEFAULT = np.array([0.8, 0.8, 0.8, 0.0, 0.0, 0.5, 1.0, 0.5, 0.0, 0.0, 1.0, 0.0, 0.03, 1.45, 0.0, 0.0, 0.0], np.float32)
MAT_INDEX = {"base_color": 0, **{k: 3 + i for i, k in enumerate(_MAT_FIELDS)}}


def material(**kw):
    """Build a 17-float material_data from keyword overrides (defaults as device_global.hpp)."""
    m = MAT_DEFAULT.copy()
    for k, v in kw.items():
        if k == "base_color":
            m[0:3] = v
        else:
            m[MAT_INDEX[k]] = v
    return m


def parse_settings(path):
    """parser.cpp:81-117 -- every key is required (nlohmann .get<> throws on a missing key)."""
    with open(path) as f:
        c = json.load(f)
    t = c["test"]
    vals = t["values"]
    test = dict(name=str(t["name"]), material_name=str(t["material_name"]), attribute_name=str(t["attribute_name"]),
                material_type=int(t["material_type"]), step_size=float(t["step_size"]),
                vec_values=[list(map(float, v)) for v in vals if isinstance(v, list)],
                flt_values=[float(v) for v in vals if not isinstance(v, list)])
    return dict(scene=str(c["scene"]), buffer_size=(int(c["buffer_size"][0]), int(c["buffer_size"][1])),
                max_path_depth=int(c["max_path_depth"]), max_samples=int(c["max_samples"]),
                environment_use=bool(c["environment_use"]), environment_auto=bool(c["environment_auto"]),
                environment_color=[float(x) for x in c["environment_color"]], environment_intensity=float(c["environment_intensity"]),
                test=test)


def parse_scene(path):
    """Returns (camera dict, materials list of (name, float32[17], texture filename or ''))."""
    with open(path) as f:
        c = json.load(f)
    cam = c["camera"]
    camera = dict(look_from=[float(x) for x in cam["look_from"]], look_at=[float(x) for x in cam["look_at"]],
                  look_up=[float(x) for x in cam["look_up"]], vertical_fov=float(cam["vertical_fov"]))
    mats = []
    for m in c["materials"]:
        data = MAT_DEFAULT.copy()
        filename = ""
        if bool(m["use_texture"]):  # parser.cpp:32-35 (KeyError if absent, as the reference throws)
            filename = m["name"] + "-textures/" + m["filename"]
        else:
            data[0:3] = [m["base_color"][0], m["base_color"][1], m["base_color"][2]]
        for i, k in enumerate(_MAT_FIELDS):
            data[3 + i] = m[k]
        mats.append((str(m["name"]), data.astype(np.float32), filename))
    return camera, mats


def _parse_index(tok, nv, nvt, nvn):
    """tinyobj triple: v, v/vt, v//vn, v/vt/vn; 1-based, negative = relative.  Missing -> -1."""
    parts = tok.split("/")

    def fix(s, n):
        if s == "":
            return -1
        i = int(s)
        return i - 1 if i > 0 else n + i

    vi = fix(parts[0], nv)
    ti = fix(parts[1], nvt) if len(parts) > 1 else -1
    ni = fix(parts[2], nvn) if len(parts) > 2 else -1
    return vi, ti, ni


def load_obj(path):
    """tinyobjloader-compatible subset with triangulate=false (mesh_loader.cpp:85-121).

    Returns a list of (shape name, mesh dict) where mesh = create_mesh(shape) below.  Shapes are
    split on 'o' and 'g' statements; faces seen before any name go to a shape named ''.
    """
    vs, vts, vns = [], [], []
    shapes = []  # (name, faces: list of list of (vi, ti, ni))
    cur_name, cur_faces = "", []

    def flush():
        nonlocal cur_faces
        if cur_faces:
            shapes.append((cur_name, cur_faces))
        cur_faces = []

    with open(path) as f:
        for line in f:
            if not line or line[0] == "#":
                continue
            tok = line.split()
            if not tok:
                continue
            k = tok[0]
            if k == "v":
                vs.append((float(tok[1]), float(tok[2]), float(tok[3])))
            elif k == "vn":
                vns.append((float(tok[1]), float(tok[2]), float(tok[3])))
            elif k == "vt":
                vts.append((float(tok[1]), float(tok[2]) if len(tok) > 2 else 0.0))
            elif k == "f":
                cur_faces.append([_parse_index(t, len(vs), len(vts), len(vns)) for t in tok[1:]])
            elif k in ("o", "g"):
                flush()
                cur_name = " ".join(tok[1:]) if len(tok) > 1 else ""
    flush()
    V = np.asarray(vs, np.float32).reshape(-1, 3)
    VT = np.asarray(vts, np.float32).reshape(-1, 2)
    VN = np.asarray(vns, np.float32).reshape(-1, 3)
    return [(name, create_mesh(faces, V, VT, VN)) for name, faces in shapes]


def create_mesh(faces, V, VT, VN):
    """mesh_loader.cpp:9-83, quirks included:
    - only 3 indices are consumed per face and the running offset advances by 3 regardless of the
      face's real vertex count (:27-81);
    - local vertices are keyed on the POSITION index only (:44-52);
    - a normal/texcoord is appended only while the attribute array is shorter than the vertex array,
      i.e. each local vertex keeps the attribute of the corner that first introduced it (and earlier
      attribute-less vertices are back-filled with it) (:55-78).
    """
    flat = [c for face in faces for c in face]
    vertex_mapping = {}
    vertices, normals, texcoords, indices = [], [], [], []
    off = 0
    for _ in range(len(faces)):
        tri = [0, 0, 0]
        for v in range(3):
            vi, ti, ni = flat[off + v]
            if vi not in vertex_mapping:
                vertex_mapping[vi] = len(vertices)
                vertices.append(V[vi])
            tri[v] = vertex_mapping[vi]
            if ni >= 0:
                while len(normals) < len(vertices):
                    normals.append(VN[ni])
            if ti >= 0:
                while len(texcoords) < len(vertices):
                    texcoords.append(VT[ti])
        indices.append(tri)
        off += 3
    return dict(vertices=np.asarray(vertices, np.float32).reshape(-1, 3), normals=np.asarray(normals, np.float32).reshape(-1, 3),
                texcoords=np.asarray(texcoords, np.float32).reshape(-1, 2), indices=np.asarray(indices, np.int32).reshape(-1, 3))


def build_entities(meshes, materials):
    """application.cpp:166-179: one entity per mesh whose name equals a material name (first match);
    material id = index in the JSON list; unmatched meshes are silently dropped."""
    ents = []
    for name, mesh in meshes:
        for pos, (mname, _, _) in enumerate(materials):
            if mname == name:
                ents.append((mesh, pos))
                break
    return ents


def flatten_scene(entities, materials, textures_by_material=None):
    """Flatten entities to one record per triangle in global order (entity order, then face order).

    textures_by_material: {material index: (H, W) uint32 RGBA8 array, row 0 = v=0 (already flipped)}.
    """
    textures_by_material = textures_by_material or {}
    pos, nrm, tcs, mi, ti = [], [], [], [], []
    textures, tex_slot = [], {}
    any_tc = False
    for mesh, mat_id in entities:
        idx = mesh["indices"]
        n = idx.shape[0]
        if mesh["normals"].shape[0] < mesh["vertices"].shape[0]:
            raise ValueError("mesh without per-vertex normals (the reference would trap, macros.hpp:5-11)")
        pos.append(mesh["vertices"][idx].reshape(n, 9))
        nrm.append(mesh["normals"][idx].reshape(n, 9))
        slot = -1
        if mat_id in textures_by_material:
            if mat_id not in tex_slot:
                tex_slot[mat_id] = len(textures)
                textures.append(np.ascontiguousarray(textures_by_material[mat_id], np.uint32))
            slot = tex_slot[mat_id]
        if mesh["texcoords"].shape[0] >= mesh["vertices"].shape[0] and mesh["vertices"].shape[0] > 0:
            tcs.append(mesh["texcoords"][idx].reshape(n, 6))
            any_tc = True
        else:
            if slot >= 0:
                raise ValueError("textured mesh without texcoords")
            tcs.append(np.zeros((n, 6), np.float32))
        mi.append(np.full(n, mat_id, np.int32))
        ti.append(np.full(n, slot, np.int32))
    cat = lambda xs, w, dt: (np.concatenate(xs).astype(dt) if xs else np.zeros((0, w) if w else (0,), dt))
    return dict(positions=cat(pos, 9, np.float32), normals=cat(nrm, 9, np.float32),
                texcoords=cat(tcs, 6, np.float32) if any_tc else None,
                material_index=cat(mi, 0, np.int32), texture_index=cat(ti, 0, np.int32),
                materials=np.stack([m for _, m, _ in materials]).astype(np.float32) if materials else np.zeros((0, MAT_FLOATS), np.float32),
                textures=textures)


def load_scene_dir(assets_dir, scene, textures_by_name=None):
    """init_program_data (application.cpp:143-181) minus settings: returns dict(camera, materials, meshes, entities)."""
    camera, materials = parse_scene(os.path.join(assets_dir, scene + ".json"))
    meshes = load_obj(os.path.join(assets_dir, scene + ".obj.scene"))
    entities = build_entities(meshes, materials)
    return dict(camera=camera, materials=materials, meshes=meshes, entities=entities)


def checker_texture(w=64, h=64, cell=8, c0=(230, 230, 230), c1=(40, 90, 200)):
    """Deterministic RGBA8 checker (stand-in for the never-committed cube-textures/cube.png)."""
    y, x = np.mgrid[0:h, 0:w]
    sel = ((x // cell) + (y // cell)) & 1
    r = np.where(sel, c1[0], c0[0]).astype(np.uint32)
    g = np.where(sel, c1[1], c0[1]).astype(np.uint32)
    b = np.where(sel, c1[2], c0[2]).astype(np.uint32)
    return (r | (g << 8) | (b << 16) | (0xFF << 24)).astype(np.uint32)
