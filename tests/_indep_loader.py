"""An INDEPENDENT reading of the reference's scene format (src/scene.rs), written from the reference's loader alone and sharing no
code with tray_rust_amd/csrc/host/scene.cpp: test infrastructure that closes the gap "oracle and product consume the same
TrayFlatScene, so a loader bug is invisible to every parity test" for the scenes the bundled JSON files describe.

It understands what cornell_box.json / smallpt.json use: film + filter, camera (fov, transform), pathtracer, the seven material
types with constant parameters, receivers / area + point emitters with sphere / disk / rectangle (plane) / mesh geometry, groups,
and `transform` lists of translate / scale / rotate_x|y|z. Everything is float64 numpy; comparisons with the product's f32
flattening use a tolerance. What it returns is a plain description: list of instances in scene order with 4x4 matrices."""
import json
import math
import os

import numpy as np


def _translate(v):
    m = np.eye(4); m[:3, 3] = v; return m


def _scale(v):
    return np.diag([v[0], v[1], v[2], 1.0])


def _rot(axis, deg):   # Transform::rotate_x / y / z (linalg/transform.rs:50-97): right-handed rotation matrices
    c, s = math.cos(math.radians(deg)), math.sin(math.radians(deg))
    m = np.eye(4)
    if axis == "x": m[1, 1], m[1, 2], m[2, 1], m[2, 2] = c, -s, s, c
    elif axis == "y": m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, s, -s, c
    else: m[0, 0], m[0, 1], m[1, 0], m[1, 1] = c, -s, s, c
    return m


def load_transform(ops):   # scene.rs:751-823: every op is multiplied from the LEFT onto what came before
    m = np.eye(4)
    for t in ops:
        ty = t["type"]
        if ty == "translate": m = _translate(t["translation"]) @ m
        elif ty == "scale":
            s = t["scaling"]
            m = _scale(s if isinstance(s, list) else [s, s, s]) @ m
        elif ty in ("rotate_x", "rotate_y", "rotate_z"): m = _rot(ty[-1], t["rotation"]) @ m
        else: raise NotImplementedError(ty)
    return m


MATERIAL_KINDS = {"matte": 0, "plastic": 1, "metal": 2, "glass": 3, "rough_glass": 4, "specular_metal": 5, "merl": 6}


def load_material(m):   # scene.rs:404-511: which JSON keys feed which parameter (c0, c1, f0, f1 of TrayMaterial)
    ty = m["type"]
    col = lambda k: list(m[k][:3])
    if ty == "matte": return dict(kind=0, c0=col("diffuse"), f0=m["roughness"])
    if ty == "plastic": return dict(kind=1, c0=col("diffuse"), c1=col("gloss"), f0=m["roughness"])
    if ty == "metal": return dict(kind=2, c0=col("refractive_index"), c1=col("absorption_coefficient"), f0=m["roughness"])
    if ty == "glass": return dict(kind=3, c0=col("reflect"), c1=col("transmit"), f0=m["eta"])
    if ty == "rough_glass": return dict(kind=4, c0=col("reflect"), c1=col("transmit"), f0=m["eta"], f1=m["roughness"])
    if ty == "specular_metal": return dict(kind=5, c0=col("refractive_index"), c1=col("absorption_coefficient"))
    raise NotImplementedError(ty)


def count_obj_triangles(path, model):   # tobj: faces of the named object, fan-triangulated
    n, cur = 0, None
    for line in open(path):
        p = line.split()
        if not p: continue
        if p[0] in ("o", "g"): cur = p[1] if len(p) > 1 else ""
        elif p[0] == "f" and cur == model: n += len(p) - 3
    return n


GEOM = {"sphere": 0, "disk": 1, "rectangle": 2, "plane": 2, "mesh": 3}


def load_objects(objs, parent, base, materials, out):   # scene.rs:513-749: depth-first, groups compose their transform on the left
    for o in objs:
        m = parent @ load_transform(o["transform"])
        if o["type"] == "group":
            load_objects(o["objects"], m, base, materials, out)
            continue
        inst = dict(name=o["name"], mat=m)
        if o["type"] == "emitter" and o["emitter"] == "point":
            inst.update(kind=2, geom_type=4, emission=o["emission"])
            out.append(inst)
            continue
        g = o["geometry"]
        gt = GEOM[g["type"]]
        params = {0: [g.get("radius")], 1: [g.get("radius"), g.get("inner_radius")], 2: [g.get("width", 2.0), g.get("height", 2.0)], 3: []}[gt]
        if g["type"] == "plane": params = [2.0, 2.0]   # scene.rs:598-600: a plane is Rectangle::new(2.0, 2.0)
        inst.update(kind=1 if o["type"] == "emitter" else 0, geom_type=gt, params=params, material=materials[o["material"]],
                    emission=o.get("emission"))
        if gt == 3: inst["triangles"] = count_obj_triangles(os.path.join(base, g["file"]), g["model"])
        out.append(inst)


def mitchell_netravali_1d(x, b, c):   # film/filter/mitchell_netravali.rs:35-48
    x = abs(2.0 * x)
    if x >= 2.0: return 0.0
    if x >= 1.0: return 1 / 6 * ((-b - 6 * c) * x ** 3 + (6 * b + 30 * c) * x ** 2 + (-12 * b - 48 * c) * x + (8 * b + 24 * c))
    return 1 / 6 * ((12 - 9 * b - 6 * c) * x ** 3 + (-18 + 12 * b + 6 * c) * x ** 2 + (6 - 2 * b))


def load(path):
    d = json.load(open(path))
    base = os.path.dirname(path)
    film = d["film"]
    flt = film["filter"]
    out = dict(width=film["width"], height=film["height"], samples=film["samples"], min_depth=d["integrator"]["min_depth"],
               max_depth=d["integrator"]["max_depth"])
    if flt["type"] == "mitchell_netravali":   # RenderTarget::new's 16x16 table at (i + 0.5) * w / 16 (render_target.rs:50-58)
        w, h, b, c = flt["width"], flt["height"], flt["b"], flt["c"]
        out["filter_table"] = np.array([[mitchell_netravali_1d((x + 0.5) * w / 16 / w, b, c) * mitchell_netravali_1d((y + 0.5) * h / 16 / h, b, c)
                                         for x in range(16)] for y in range(16)])
    materials = {m["name"]: load_material(m) for m in d["materials"]}
    cam = d["camera"] if "camera" in d else d["cameras"][0]
    out["cam_world"] = load_transform(cam["transform"])
    out["fov"] = cam["fov"]
    insts = []
    load_objects(d["objects"], np.eye(4), base, materials, insts)
    out["instances"] = insts
    out["lights"] = [i for i, x in enumerate(insts) if x["kind"] != 0]
    return out


# ---- assets: OBJ meshes (tobj 0.1.6 as called by mesh.rs:50-78), MERL tables (material/merl.rs:51-84) ----------------------
def read_obj_model(path, model):
    """Triangles of the object / group `model` of an OBJ file as tobj hands them to Mesh::new: faces fan-triangulated, every face
    corner v/vt/vn resolved to (position, texcoord, normal). Returns (positions [n,3,3], normals [n,3,3], texcoords [n,3,2]) in file
    order, float64 as written in the file."""
    v, vt, vn = [], [], []
    tris_p, tris_n, tris_t = [], [], []
    cur = None
    for line in open(path):
        p = line.split()
        if not p or p[0].startswith("#"):
            continue
        if p[0] == "v": v.append([float(x) for x in p[1:4]])
        elif p[0] == "vt": vt.append([float(x) for x in p[1:3]])
        elif p[0] == "vn": vn.append([float(x) for x in p[1:4]])
        elif p[0] in ("o", "g"): cur = p[1] if len(p) > 1 else ""
        elif p[0] == "f" and cur == model:
            corners = []
            for tok in p[1:]:
                a = tok.split("/")
                idx = [int(x) if x else 0 for x in a] + [0] * (3 - len(a))
                res = []
                for k, arr in zip(idx, (v, vt, vn)):   # OBJ indices are 1-based, negative = relative to the end
                    res.append(arr[k - 1] if k > 0 else arr[k])
                corners.append(res)
            for i in range(1, len(corners) - 1):
                tri = (corners[0], corners[i], corners[i + 1])
                tris_p.append([c[0] for c in tri]); tris_t.append([c[1] for c in tri]); tris_n.append([c[2] for c in tri])
    return np.array(tris_p, np.float64), np.array(tris_n, np.float64), np.array(tris_t, np.float64)


def read_merl(path):
    """Merl::load_file (material/merl.rs:51-84): i32 x 3 header (90, 90, 180), then the red, green and blue planes as f64, scaled by
    (1, 1, 1.66) / 1500 -- green WITHOUT MERL's own 1.15 (quirk Q10) --, narrowed to f32, clamped at 0, interleaved rgb."""
    raw = open(path, "rb").read()
    dims = np.frombuffer(raw, "<i4", 3)
    assert tuple(dims) == (90, 90, 180), dims
    n = int(np.prod(dims))
    planes = np.frombuffer(raw, "<f8", 3 * n, offset=12).reshape(3, n)
    scale = np.array([1.0 / 1500.0, 1.0 / 1500.0, 1.66 / 1500.0])
    out = np.maximum(np.float32(0.0), (planes * scale[:, None]).astype(np.float32))
    return np.ascontiguousarray(out.T).reshape(-1)   # brdf[3 * i + c]


# ---- keyframed transforms (scene.rs:825-850, animated_transform.rs:22-86): the stack of spline levels of every instance --------
def _level(o):
    """one level of an AnimatedTransform: a constant (`transform`) or a B-spline over control transforms (`keyframes`)"""
    if "keyframes" in o:
        k = o["keyframes"]
        return dict(mats=[load_transform(c["transform"]) for c in k["control_points"]], knots=[float(x) for x in k["knots"]], degree=int(k.get("degree", 3)))
    return dict(mats=[load_transform(o.get("transform", []))], knots=None, degree=0)


def instance_stacks(objs, parents=()):
    """[(name, [own level, parent group's level, grandparent's, ...])] in scene order: a group composes as `group * child`, which
    APPENDS the group's levels behind the child's (animated_transform.rs:78-86, scene.rs:568-573)"""
    out = []
    for o in objs:
        if o["type"] == "group":
            out += instance_stacks(o["objects"], (_level(o),) + tuple(parents))
        else:
            out.append((o["name"], [_level(o)] + list(parents)))
    return out


def keyframe_matrix(translation, rotation, scaling):
    """Keyframe::transform (keyframe.rs:60-63): T * R(q) * S, q = (x, y, z, w) (quaternion.rs:67-99)"""
    x, y, z, w = rotation
    r = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    m = np.eye(4)
    m[:3, :3] = r @ np.diag(scaling)
    m[:3, 3] = translation
    return m


def keyframe_of(m):
    """Keyframe::new / decompose (keyframe.rs:22-58) of a 4x4 transform, with numpy's SVD where the reference calls la::SVD: the
    translation column; q = U V^T (negated together with p when det < 0); p = V S V^T, of which ONLY THE DIAGONAL is kept as the
    scaling -- a control transform whose stretch is not axis-aligned after the rotation is therefore not reproduced by T R S (the
    reference's behaviour). Returns the 4x4 matrix T * q * diag(p) the keyframe stands for."""
    a = np.asarray(m, np.float64)[:3, :3]
    u, sv, vt = np.linalg.svd(a)
    q = u @ vt
    p = vt.T @ np.diag(sv) @ vt
    if np.linalg.det(q) < 0.0:
        q, p = -q, -p
    out = np.eye(4)
    out[:3, :3] = q @ np.diag(np.diag(p))
    out[:3, 3] = np.asarray(m, np.float64)[:3, 3]
    return out
