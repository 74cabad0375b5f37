"""Benchmark scene descriptions in the reference's JSON schema.

The reference ships its scenes as JSON under /root/reference/scenes; that directory does not exist
on the GPU box, and resolution / spp can only be changed inside the JSON (src/scene.rs:185-206).
These builders emit the same scenes (all numbers are the reference's: scenes/cornell_box.json,
scenes/smallpt.json, scenes/models/cube.obj) with the film block parameterised, so BASELINE.json's
configs C1-C3 are one call each:

    cornell_box(400, 400, 64)      C1      cornell_box(1920, 1080, 1024)   C2
    smallpt(1920, 1080, 4096)      C3

`write_assets(dir)` writes <dir>/cornell_box.json, <dir>/smallpt.json and <dir>/models/cube.obj.
"""
import json
import os

_FILTER = {"type": "mitchell_netravali", "width": 2.0, "height": 2.0, "b": 0.333333333333333333, "c": 0.333333333333333333}


def _film(width, height, samples):
    return {"width": int(width), "height": int(height), "samples": int(samples), "frames": 1, "start_frame": 0,
            "end_frame": 0, "scene_time": 0, "filter": dict(_FILTER)}


_CAMERA = {"fov": 30, "transform": [{"type": "translate", "translation": [0, 12, -60]}]}
_INTEGRATOR = {"type": "pathtracer", "min_depth": 4, "max_depth": 8}


def _plane(name, material, transform):
    return {"name": name, "type": "receiver", "material": material, "geometry": {"type": "plane"}, "transform": transform}


def _t(x, y, z):
    return {"type": "translate", "translation": [x, y, z]}


def _s(v):
    return {"type": "scale", "scaling": v}


def _rx(a):
    return {"type": "rotate_x", "rotation": a}


def _ry(a):
    return {"type": "rotate_y", "rotation": a}


def cornell_box(width=800, height=600, samples=4):
    """scenes/cornell_box.json: 5 rectangle walls in a group, a 6x6 rectangle area light, two cube meshes."""
    walls = [
        _plane("back_wall", "white_wall", [_s([15, 12, 1]), _t(0, 0, 20)]),
        _plane("left_wall", "red_wall", [_s([20, 12, 1]), _ry(90.0), _t(-15.0, 0, 0)]),
        _plane("right_wall", "green_wall", [_s([20, 12, 1]), _ry(-90.0), _t(15.0, 0, 0)]),
        _plane("top_wall", "white_wall", [_s([15, 20, 1]), _rx(90.0), _t(0.0, 12, 0)]),
        _plane("bottom_wall", "white_wall", [_s([15, 20, 1]), _rx(90), _t(0.0, -12, 0)]),
    ]
    cube = {"type": "mesh", "file": "models/cube.obj", "model": "Cube"}
    return {
        "film": _film(width, height, samples),
        "camera": dict(_CAMERA),
        "integrator": dict(_INTEGRATOR),
        "materials": [
            {"type": "matte", "name": "white_wall", "diffuse": [0.740063, 0.742313, 0.733934], "roughness": 1.0},
            {"type": "matte", "name": "red_wall", "diffuse": [0.366046, 0.0371827, 0.0416385], "roughness": 1.0},
            {"type": "matte", "name": "green_wall", "diffuse": [0.162928, 0.408903, 0.0833759], "roughness": 1.0},
            {"type": "plastic", "name": "white_plastic", "diffuse": [0.8, 0.8, 0.8], "gloss": [0.6, 0.6, 0.6], "roughness": 0.5},
        ],
        "objects": [
            {"type": "group", "name": "walls", "transform": [_t(0, 12, 0)], "objects": walls},
            {"name": "light", "type": "emitter", "material": "white_wall", "emitter": "area",
             "emission": [1, 0.772549, 0.560784, 40], "geometry": {"type": "rectangle", "width": 6, "height": 6},
             "transform": [_rx(90), _t(0, 23.8, 0)]},
            {"name": "tall_cube", "type": "receiver", "material": "white_plastic", "geometry": dict(cube),
             "transform": [_s([4, 10, 4]), _ry(-20), _t(-6, 5, 6)]},
            {"name": "short_block", "type": "receiver", "material": "white_plastic", "geometry": dict(cube),
             "transform": [_s([4, 5, 4]), _ry(15), _t(4, 2.5, -3.0)]},
        ],
    }


def smallpt(width=800, height=600, samples=16):
    """scenes/smallpt.json: 5 rectangle walls, a Beckmann metal sphere, a glass sphere, a sphere light."""
    walls = [
        _plane("back_wall", "white_wall", [_s(32.0), _t(0, 0, 20)]),
        _plane("left_wall", "red_wall", [_s(32), _ry(90.0), _t(-15.0, 0, 0)]),
        _plane("right_wall", "blue_wall", [_s(32.0), _ry(-90.0), _t(15.0, 0, 0)]),
        _plane("top_wall", "white_wall", [_s(32.0), _rx(90.0), _t(0.0, 12, 0)]),
        _plane("bottom_wall", "white_wall", [_s(32.0), _rx(90), _t(0.0, -12, 0)]),
    ]
    sphere = {"type": "sphere", "radius": 1.0}
    return {
        "film": _film(width, height, samples),
        "camera": dict(_CAMERA),
        "integrator": dict(_INTEGRATOR),
        "materials": [
            {"type": "matte", "name": "white_wall", "diffuse": [1.0, 1.0, 1.0], "roughness": 1.0},
            {"type": "matte", "name": "red_wall", "diffuse": [1.0, 0.2, 0.2], "roughness": 1.0},
            {"type": "matte", "name": "blue_wall", "diffuse": [0.2, 0.2, 1.0], "roughness": 1.0},
            {"type": "metal", "name": "metal", "refractive_index": [0.155265, 0.116723, 0.138381],
             "absorption_coefficient": [4.82835, 3.12225, 2.14696], "roughness": 0.2},
            {"type": "plastic", "name": "plastic", "gloss": [0.8, 0.8, 0.8], "diffuse": [0.8, 0.2, 0.2], "roughness": 0.02},
            {"type": "glass", "name": "glass", "reflect": [1.0, 1.0, 1.0], "transmit": [1.0, 1.0, 1.0], "eta": 1.52},
        ],
        "objects": [
            {"type": "group", "name": "walls", "transform": [_t(0, 12, 0)], "objects": walls},
            {"name": "metal_sphere", "type": "receiver", "material": "metal", "geometry": dict(sphere),
             "transform": [_s(5.0), _t(-6.0, 5.0, 8.0)]},
            {"name": "glass_sphere", "type": "receiver", "material": "glass", "geometry": dict(sphere),
             "transform": [_s(5.0), _t(6.0, 5.0, -2.0)]},
            {"name": "light", "type": "emitter", "material": "white_wall", "emitter": "area",
             "emission": [0.780131, 0.780409, 0.775833, 60], "geometry": dict(sphere), "transform": [_t(0.0, 22, 0)]},
        ],
    }


# scenes/models/cube.obj (Blender export): 8 positions, 22 texcoords, 6 normals, 6 quads as (v, vt, vn)
_CUBE_V = [(1.0, -1.0, -1.0), (1.0, -1.0, 1.0), (-1.0, -1.0, 1.0), (-1.0, -1.0, -1.0),
           (1.0, 1.0, -0.999999), (0.999999, 1.0, 1.000001), (-1.0, 1.0, 1.0), (-1.0, 1.0, -1.0)]
_CUBE_VT = [(0.0, 0.334353), (0.332314, 0.333333), (0.333333, 0.665647), (0.001019, 0.666667), (1.0, 0.001019),
            (0.998981, 0.333333), (0.666667, 0.332314), (0.667686, 0.0), (1.0, 0.665647), (0.667686, 0.666667),
            (0.666667, 0.334353), (0.334353, 0.666667), (0.333333, 0.334353), (0.665647, 0.333333), (0.666667, 0.665647),
            (0.333333, 0.332314), (0.00102, 0.333333), (0.0, 0.00102), (0.332314, 0.0), (0.333333, 0.001019),
            (0.665647, 0.0), (0.334353, 0.333333)]
_CUBE_VN = [(0.0, -1.0, 0.0), (0.0, 1.0, 0.0), (1.0, 0.0, 0.0), (-0.0, -0.0, 1.0), (-1.0, -0.0, -0.0), (0.0, 0.0, -1.0)]
_CUBE_F = [[(1, 1, 1), (2, 2, 1), (3, 3, 1), (4, 4, 1)], [(5, 5, 2), (8, 6, 2), (7, 7, 2), (6, 8, 2)],
           [(1, 6, 3), (5, 9, 3), (6, 10, 3), (2, 11, 3)], [(2, 12, 4), (6, 13, 4), (7, 14, 4), (3, 15, 4)],
           [(3, 16, 5), (7, 17, 5), (8, 18, 5), (4, 19, 5)], [(5, 20, 6), (1, 21, 6), (4, 7, 6), (8, 22, 6)]]


def cube_obj():
    lines = ["o Cube"]
    lines += ["v %.6f %.6f %.6f" % v for v in _CUBE_V]
    lines += ["vt %.6f %.6f" % t for t in _CUBE_VT]
    lines += ["vn %.6f %.6f %.6f" % n for n in _CUBE_VN]
    lines.append("s off")
    lines += ["f " + " ".join("%d/%d/%d" % c for c in face) for face in _CUBE_F]
    return "\n".join(lines) + "\n"


# ---- C4 stand-ins: the reference's models/dragon.obj and brdfs/*.binary are not distributed with it.
def dragon_scene(width=1920, height=1080, samples=2048, material="merl"):
    """BASELINE.json configs[3] stand-in: cornell walls + disk light (scenes/logo_with_friends.json:186-199) + ONE
    ~871k-triangle mesh named 'dragon' with the transform of logo_with_friends.json:260-283 and a MERL material."""
    d = cornell_box(width, height, samples)
    d["camera"] = {"fov": 28, "transform": [_t(0.0, 12, -52)]}
    mats = [m for m in d["materials"] if m["type"] == "matte"]
    if material == "merl":
        mats.append({"name": "blue_acrylic", "type": "merl", "file": "brdfs/blue-acrylic.binary"})
    else:
        mats.append({"type": "plastic", "name": "blue_acrylic", "diffuse": [0.1, 0.2, 0.7], "gloss": [0.6, 0.6, 0.6], "roughness": 0.1})
    d["materials"] = mats
    walls = d["objects"][0]
    light = {"name": "light", "type": "emitter", "material": "white_wall", "emitter": "area", "emission": [1, 0.772549, 0.560784, 40],
             "geometry": {"type": "disk", "radius": 3.5, "inner_radius": 0.0}, "transform": [_rx(90), _t(0, 23.8, 0)]}
    dragon = {"name": "dragon", "type": "receiver", "material": "blue_acrylic",
              "geometry": {"type": "mesh", "file": "models/dragon.obj", "model": "dragon"},
              "transform": [_s(13.0), _ry(55), _t(8.5, 3.7, 1.5)]}
    d["objects"] = [walls, light, dragon]
    return d


def knot_mesh(grid_u, grid_v=None, seed=7, p=2, q=3, tube=0.22, noise=0.25, extent=1.0):
    """Closed torus-topology quad grid: a (p,q) torus-knot tube with 4-octave value-noise radial displacement, smooth
    normals, uv = grid coordinates, longest bounding-box side = extent, y up. Returns (pos, uv, normals, quads) with
    1-based quad indices local to this mesh; 2 * grid_u * grid_v triangles."""
    import numpy as np
    nu = int(grid_u); nv = int(grid_v or grid_u)
    u = (np.arange(nu) / nu)[:, None] * 2 * np.pi          # along the knot
    v = (np.arange(nv) / nv)[None, :] * 2 * np.pi          # around the tube

    def knot(t):
        r = 0.5 * (2 + np.cos(q * t))
        return np.stack([r * np.cos(p * t), r * np.sin(p * t), -0.5 * np.sin(q * t)], -1)
    c = knot(u[:, 0])
    dt = 1e-4
    tangent = knot(u[:, 0] + dt) - knot(u[:, 0] - dt)
    tangent /= np.linalg.norm(tangent, axis=1, keepdims=True)
    ref = np.array([0.0, 0.0, 1.0])
    nrm = np.cross(tangent, ref); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    bin_ = np.cross(tangent, nrm)
    rng = np.random.default_rng(seed)
    disp = np.zeros((nu, nv))
    for octave in range(4):   # periodic value noise
        k = 8 << octave
        g = rng.uniform(-1, 1, (k, k))
        iu = (np.arange(nu) * k / nu); iv = (np.arange(nv) * k / nv)
        u0 = np.floor(iu).astype(int); v0 = np.floor(iv).astype(int)
        fu = (iu - u0)[:, None]; fv = (iv - v0)[None, :]
        fu = fu * fu * (3 - 2 * fu); fv = fv * fv * (3 - 2 * fv)
        a = g[u0 % k][:, v0 % k]; b = g[(u0 + 1) % k][:, v0 % k]; cc = g[u0 % k][:, (v0 + 1) % k]; dd = g[(u0 + 1) % k][:, (v0 + 1) % k]
        disp += (a * (1 - fu) * (1 - fv) + b * fu * (1 - fv) + cc * (1 - fu) * fv + dd * fu * fv) / (2 ** octave)
    radius = tube * (1 + noise * disp)
    pos = c[:, None, :] + radius[..., None] * (np.cos(v)[..., None] * nrm[:, None, :] + np.sin(v)[..., None] * bin_[:, None, :])
    # smooth normals from the periodic grid
    du = np.roll(pos, -1, 0) - np.roll(pos, 1, 0); dv = np.roll(pos, -1, 1) - np.roll(pos, 1, 1)
    nn = np.cross(dv, du); nn /= np.linalg.norm(nn, axis=2, keepdims=True)
    lo, hi = pos.reshape(-1, 3).min(0), pos.reshape(-1, 3).max(0)
    pos = (pos - (lo + hi) / 2) * (extent / (hi - lo).max())
    pos = pos[..., [0, 2, 1]]; nn = nn[..., [0, 2, 1]]          # stand it up: y is up in the scene
    uv = np.stack(np.meshgrid(np.arange(nu) / nu, np.arange(nv) / nv, indexing="ij"), -1)
    idx = (np.arange(nu)[:, None] * nv + np.arange(nv)[None, :])
    a = idx; b = np.roll(idx, -1, 0); cc = np.roll(np.roll(idx, -1, 0), -1, 1); dd = np.roll(idx, -1, 1)
    quads = np.stack([a, b, cc, dd], -1).reshape(-1, 4) + 1
    return pos.reshape(-1, 3), uv.reshape(-1, 2), nn.reshape(-1, 3), quads


def write_obj(path, objects):
    """objects: [(name, pos, uv, normals, quads)] -> a Wavefront OBJ with one `o` block per object (v/vt/vn, quad faces,
    indices global across the file as the format requires). Returns the triangle count."""
    import numpy as np
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    tris = 0
    base = 0
    with open(path, "w") as f:
        for name, pos, uv, nn, quads in objects:
            f.write("o %s\n" % name)
            np.savetxt(f, pos, fmt="v %.7f %.7f %.7f")
            np.savetxt(f, uv, fmt="vt %.6f %.6f")
            np.savetxt(f, nn, fmt="vn %.6f %.6f %.6f")
            q = quads + base
            np.savetxt(f, np.stack([q[:, 0]] * 3 + [q[:, 1]] * 3 + [q[:, 2]] * 3 + [q[:, 3]] * 3, -1), fmt="f %d/%d/%d %d/%d/%d %d/%d/%d %d/%d/%d")
            base += len(pos)
            tris += 2 * len(quads)
    return tris


def write_dragon_obj(path, grid=660, seed=7, extent=0.2):
    """Deterministic stand-in for the Stanford dragon: knot_mesh on a grid x grid quad mesh (grid=660 -> 871 200 triangles /
    435 600 vertices; the real dragon has 871 414 / 437 645), written as a real OBJ with v/vt/vn (both required,
    src/geometry/mesh.rs:57-61). The longest bounding-box side is `extent` (the scanned dragon is about 0.2 units; the
    scene scales it by 13)."""
    return write_obj(path, [("dragon",) + knot_mesh(grid, grid, seed=seed, extent=extent)])


def write_merl_binary(path, kd=(0.05, 0.12, 0.45), ks=0.35, alpha=0.08):
    """MERL-format file (3 x i32 header 90,90,180 + three planes of f64, material/merl.rs:51-84) filled with an analytic
    diffuse + Cook-Torrance lobe evaluated at the bin centres and divided by the loader's channel scales."""
    import numpy as np
    nth, ntd, npd = 90, 90, 180
    th = ((np.arange(nth) + 0.5) / nth) ** 2 * (np.pi / 2)            # theta_h bins are sqrt-spaced (bxdf/merl.rs:76)
    td = (np.arange(ntd) + 0.5) / ntd * (np.pi / 2)
    pd = (np.arange(npd) + 0.5) / npd * np.pi
    TH, TD, PD = np.meshgrid(th, td, pd, indexing="ij")
    # half-vector frame -> cos(theta_i), cos(theta_o)
    wd = np.stack([np.sin(TD) * np.cos(PD), np.sin(TD) * np.sin(PD), np.cos(TD)], -1)
    ct, st = np.cos(TH), np.sin(TH)
    wi = np.stack([ct * wd[..., 0] + st * wd[..., 2], wd[..., 1], -st * wd[..., 0] + ct * wd[..., 2]], -1)
    wo = wi * np.array([-1.0, -1.0, 1.0]) + 0  # mirror of wd about h, rotated the same way
    wdm = wd * np.array([-1.0, -1.0, 1.0])
    wo = np.stack([ct * wdm[..., 0] + st * wdm[..., 2], wdm[..., 1], -st * wdm[..., 0] + ct * wdm[..., 2]], -1)
    ci, co = np.clip(wi[..., 2], 1e-3, 1), np.clip(wo[..., 2], 1e-3, 1)
    d = np.exp(-(np.tan(TH) ** 2) / alpha ** 2) / (np.pi * alpha ** 2 * np.cos(TH) ** 4 + 1e-12)
    fres = 0.04 + 0.96 * (1 - np.cos(TD)) ** 5
    spec = ks * d * fres / (4 * ci * co)
    scales = (1.0 / 1500.0, 1.0 / 1500.0, 1.66 / 1500.0)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        f.write(np.array([nth, ntd, npd], dtype="<i4").tobytes())
        for ch in range(3):
            val = (kd[ch] / np.pi + spec) / scales[ch]
            f.write(val.astype("<f8").tobytes())


def write_dragon_assets(directory, film=(1920, 1080, 2048), grid=660, material="merl", extent=0.2):
    """models/dragon.obj + brdfs/blue-acrylic.binary + dragon.json under `directory`."""
    n = write_dragon_obj(os.path.join(directory, "models", "dragon.obj"), grid, extent=extent)
    if material == "merl":
        write_merl_binary(os.path.join(directory, "brdfs", "blue-acrylic.binary"))
    os.makedirs(os.path.join(directory, "models"), exist_ok=True)
    with open(os.path.join(directory, "models", "cube.obj"), "w") as f:
        f.write(cube_obj())
    return write_scene(dragon_scene(*film, material=material), os.path.join(directory, "dragon.json")), n


# ---- moving scenes (BASELINE.json configs[4] is scenes/tr15.json, whose 25 models and 5 BRDFs are not
# distributed). These builders use the same JSON features tr15.json uses: "keyframes" (cubic B-splines of TRS control
# points) on the camera, on objects and on groups, animated "emission" key lists, shutter_size, frames / scene_time.
def _key(*transform):
    return {"transform": list(transform)}


def _clamped_knots(n_points, t0, t1, degree=3):
    inner = n_points - degree - 1
    return [t0] * (degree + 1) + [t0 + (t1 - t0) * (i + 1) / (inner + 1) for i in range(inner)] + [t1] * (degree + 1)


def moving_box(width=160, height=120, samples=16, frames=8, scene_time=2.0, shutter_size=0.5):
    """Cornell walls + every kind of motion the reference supports: a camera on a cubic spline, a sphere that translates,
    rotates and scales, a cube mesh whose GROUP moves (static child under a moving parent), a disk light that slides and
    changes colour, a point light fading in."""
    d = cornell_box(width, height, samples)
    d["film"].update({"frames": frames, "start_frame": 0, "end_frame": frames - 1, "scene_time": scene_time})
    d["camera"] = {"fov": 30, "shutter_size": shutter_size, "keyframes": {
        "control_points": [_key(_t(0, 12, -60)), _key(_ry(-2), _t(2, 13, -58)), _key(_ry(-5), _t(5, 14, -55)),
                           _key(_ry(-7), _t(7, 13, -53)), _key(_ry(-8), _t(8, 12, -52))],
        "knots": _clamped_knots(5, 0.0, scene_time)}}
    d["materials"] += [{"type": "metal", "name": "metal", "refractive_index": [0.155265, 0.116723, 0.138381],
                        "absorption_coefficient": [4.82835, 3.12225, 2.14696], "roughness": 0.2},
                       {"type": "glass", "name": "glass", "reflect": [1, 1, 1], "transmit": [1, 1, 1], "eta": 1.52}]
    walls = d["objects"][0]
    cube = {"type": "mesh", "file": "models/cube.obj", "model": "Cube"}
    ball = {"name": "ball", "type": "receiver", "material": "metal", "geometry": {"type": "sphere", "radius": 1.0},
            "keyframes": {"control_points": [_key(_s(3.0), _t(-8, 3, 4)), _key(_s(3.5), _rx(40), _t(-4, 9, 2)),
                                             _key(_s([4.0, 3.0, 4.0]), _rx(90), _ry(45), _t(2, 6, 0)), _key(_s(3.0), _rx(170), _t(7, 3, -2))],
                          "knots": _clamped_knots(4, 0.0, scene_time)}}
    spinner = {"type": "group", "name": "spinner",
               "keyframes": {"control_points": [_key(_t(6, 0, 8)), _key(_ry(60), _t(5, 1, 7)), _key(_ry(120), _t(4, 2, 6)),
                                                _key(_ry(180), _t(3, 1, 5)), _key(_ry(240), _t(2, 0, 4))],
                             "knots": _clamped_knots(5, 0.0, scene_time)},
               "objects": [{"name": "block", "type": "receiver", "material": "white_plastic", "geometry": dict(cube),
                            "transform": [_s([3, 5, 3]), _t(0, 5, 0)]},
                           {"name": "lens", "type": "receiver", "material": "glass", "geometry": {"type": "sphere", "radius": 2.0},
                            "transform": [_t(0, 12.5, 0)]}]}
    lamp = {"name": "light", "type": "emitter", "material": "white_wall", "emitter": "area",
            "emission": [{"time": 0.0, "color": [1, 0.772549, 0.560784, 30]}, {"time": scene_time * 0.5, "color": [0.6, 0.8, 1.0, 55]},
                         {"time": scene_time, "color": [1, 0.5, 0.4, 40]}],
            "geometry": {"type": "disk", "radius": 3.5, "inner_radius": 0.0},
            "keyframes": {"control_points": [_key(_rx(90), _t(-6, 23.8, 0)), _key(_rx(90), _t(-2, 23.8, 3)),
                                             _key(_rx(90), _t(3, 23.8, 1)), _key(_rx(90), _t(6, 23.8, -2))],
                          "knots": _clamped_knots(4, 0.0, scene_time)}}
    spark = {"name": "spark", "type": "emitter", "emitter": "point",
             "emission": [{"time": 0.0, "color": [1, 0.9, 0.8, 0]}, {"time": scene_time, "color": [1, 0.9, 0.8, 400]}],
             "transform": [_t(-10, 18, -6)]}
    d["objects"] = [walls, lamp, spark, ball, spinner]
    return d


def write_moving_box(directory, **kw):
    os.makedirs(os.path.join(directory, "models"), exist_ok=True)
    with open(os.path.join(directory, "models", "cube.obj"), "w") as f:
        f.write(cube_obj())
    return write_scene(moving_box(**kw), os.path.join(directory, "moving_box.json"))


# C5 stand-in. scenes/tr15.json needs 13 OBJ files (25 models) and 5 MERL tables that are not distributed, and the file itself
# is not available on the GPU box, so the benchmark scene is generated: same film / integrator settings, the same counts
# (59 instances in 25 meshes, 10 keyed lights, 14 moving instances, a 12-point camera spline, 20 materials of the same
# types) and mesh sizes of the same order, with its own choreography. tests/test_tr15_compat.py loads the real tr15.json.
_TR15_MODELS = {   # file -> [(model, grid_u, grid_v)] at detail = 1 (2 * gu * gv triangles each)
    "models/cone.obj": [("Cone", 8, 4)],
    "models/teapot2.obj": [("Base", 44, 36), ("Top", 28, 22)],
    "models/u_logo.obj": [("U_Logo", 100, 100)],
    "models/buddha.obj": [("buddha", 737, 737)],
    "models/dragon.obj": [("dragon", 660, 660)],
    "models/kenny_nl/Tree_01.obj": [("Leaves", 36, 28), ("Trunk", 16, 12)],
    "models/kenny_nl/Tree_02.obj": [("Leaves", 40, 30), ("Trunk", 16, 12)],
    "models/rust_logo.obj": [("rust_logo", 174, 174)],
    "models/lucy.obj": [("lucy", 512, 512)],
    "models/teapot.obj": [("Teapot", 56, 56)],
    "models/ajax.obj": [("Ajax", 522, 522)],
    "models/cow.obj": [("Cow", 54, 54)],
    "models/kenny_nl/tree_1_ornamented.obj": [("Leaves", 48, 36), ("Trunk", 16, 12), ("Tinsel", 60, 8), ("Sphere1", 16, 12), ("Sphere2", 16, 12),
                                              ("Sphere3", 16, 12), ("Sphere4", 16, 12), ("Sphere5", 16, 12), ("Bunny", 60, 48), ("Suzanne", 32, 24)],
}
_TR15_BRDFS = {"oxidized_steel": ("brdfs/black-oxidized-steel.binary", (0.02, 0.02, 0.025), 0.25, 0.12),
               "silver_paint": ("brdfs/silver-paint.binary", (0.25, 0.25, 0.27), 0.5, 0.2),
               "gold_metallic_paint": ("brdfs/gold-metallic-paint.binary", (0.35, 0.22, 0.05), 0.6, 0.15),
               "blue_acrylic": ("brdfs/blue-acrylic.binary", (0.05, 0.12, 0.45), 0.35, 0.08),
               "brass": ("brdfs/brass.binary", (0.3, 0.2, 0.06), 0.8, 0.1)}


def tr15_like(width=1920, height=1080, samples=512, frames=600, scene_time=25.0):
    import math
    mesh = lambda f, m: {"type": "mesh", "file": f, "model": m}
    recv = lambda name, material, geom, transform=None, keyframes=None: dict(
        {"name": name, "type": "receiver", "material": material, "geometry": geom},
        **({"keyframes": keyframes} if keyframes else {"transform": transform}))
    disk = lambda r: {"type": "disk", "radius": r, "inner_radius": 0.0}
    warm = [1, 0.772549, 0.560784]

    def light(name, radius, keys, transform):
        return {"name": name, "type": "emitter", "material": "white_wall", "emitter": "area", "geometry": disk(radius),
                "emission": [{"time": t, "color": warm[:2] + [b, e]} for (t, b, e) in keys], "transform": transform}

    def spline(points, t0, t1):
        return {"control_points": [_key(*p) for p in points], "knots": _clamped_knots(len(points), t0, t1)}

    cam_pts = []
    for i in range(12):
        a = -30.0 + 80.0 * i / 11.0
        r = 58.0 - 16.0 * i / 11.0
        h = 12.0 + 6.0 * math.sin(math.pi * i / 11.0)
        cam_pts.append((_rx(4.0), _ry(a), _t(-r * math.sin(math.radians(a)), h, -r * math.cos(math.radians(a)))))
    materials = [
        {"type": "matte", "name": "white_wall", "diffuse": [0.740063, 0.742313, 0.733934], "roughness": 1.0},
        {"type": "matte", "name": "red_wall", "diffuse": [0.366046, 0.0371827, 0.0416385], "roughness": 1.0},
        {"type": "matte", "name": "green_wall", "diffuse": [0.162928, 0.408903, 0.0833759], "roughness": 1.0},
        {"type": "matte", "name": "blue_sky", "diffuse": [0.3, 0.5, 0.8], "roughness": 0.0},
        {"type": "matte", "name": "pantone_yellow", "diffuse": [0.9, 0.75, 0.1], "roughness": 0.5},
        {"type": "matte", "name": "sienna_matte", "diffuse": [0.53, 0.32, 0.18], "roughness": 0.8},
        {"type": "matte", "name": "forest_green_matte", "diffuse": [0.13, 0.45, 0.13], "roughness": 0.8},
        {"type": "plastic", "name": "white_plastic", "diffuse": [0.8, 0.8, 0.8], "gloss": [0.6, 0.6, 0.6], "roughness": 0.5},
        {"type": "plastic", "name": "green_plastic", "diffuse": [0.1, 0.7, 0.2], "gloss": [0.6, 0.6, 0.6], "roughness": 0.2},
        {"type": "plastic", "name": "u_logo_plastic", "diffuse": [0.8, 0.05, 0.05], "gloss": [0.5, 0.5, 0.5], "roughness": 0.3},
        {"type": "plastic", "name": "u_logo_plastic_shiny", "diffuse": [0.8, 0.05, 0.05], "gloss": [0.8, 0.8, 0.8], "roughness": 0.05},
        {"type": "plastic", "name": "pantone_yellow_plastic", "diffuse": [0.9, 0.75, 0.1], "gloss": [0.6, 0.6, 0.6], "roughness": 0.1},
        {"type": "glass", "name": "glass", "reflect": [1, 1, 1], "transmit": [1, 1, 1], "eta": 1.52},
        {"type": "metal", "name": "silver_metal", "refractive_index": [0.155265, 0.116723, 0.138381], "absorption_coefficient": [4.82835, 3.12225, 2.14696], "roughness": 0.1},
        {"type": "metal", "name": "rough_silver_metal", "refractive_index": [0.155265, 0.116723, 0.138381], "absorption_coefficient": [4.82835, 3.12225, 2.14696], "roughness": 0.3},
    ] + [{"type": "merl", "name": n, "file": f} for n, (f, _, _, _) in _TR15_BRDFS.items()]

    def ring(i, n, r):   # i-th of n spots on a circle of radius r on the stage
        a = 2 * math.pi * i / n
        return r * math.sin(a), r * math.cos(a)

    objects = []
    # stage: three walls that slide in (own spline + the group's spline), floor, backdrop, sky dome
    def wall(name, material, rot, pos):
        pts = [(_s([64, 30, 1]), rot, _t(pos[0], pos[1] + dy, pos[2])) for dy in (60.0, 35.0, 10.0, 0.0)]
        return recv(name, material, {"type": "plane"}, keyframes=spline(pts, 2.5, 6.0))
    objects.append({"type": "group", "name": "walls", "keyframes": spline([(_t(0, 0, 30 - 10 * k),) for k in range(4)], 2.5, 6.0),
                    "objects": [wall("back_wall", "white_wall", _ry(0), (0, 30, 64)), wall("left_wall", "red_wall", _ry(90), (-64, 30, 0)),
                                wall("right_wall", "green_wall", _ry(-90), (64, 30, 0))]})
    objects.append(recv("floor", "white_plastic", {"type": "plane"}, [_s(120), _rx(90), _t(0, 0, 0)]))
    objects.append(recv("far_backdrop", "blue_sky", {"type": "plane"}, [_s(200), _t(0, 0, 150)]))
    objects.append(recv("sky_dome", "blue_sky", {"type": "sphere", "radius": 300.0}, [_t(0, 0, 0)]))
    # four spot lights on moving arms
    cone_knots = [(0, 9), (0, 8), (0, 8), (0, 8)]
    for i in range(4):
        x, z = ring(i, 4, 22.0)
        arm = [(_ry(40.0 * k), _t(x * (1 - 0.15 * k), 30 - 2 * k, z * (1 - 0.15 * k))) for k in range(6 - (i > 1) - (i > 2))]
        objects.append({"type": "group", "name": "light_cone%d" % (i + 1), "keyframes": spline(arm, *cone_knots[i]), "objects": [
            light("cone_light", 2.0, [(0, 0.560784, 0), (2 + i, 0.560784, 0), (4 + i, 0.560784, 180)], [_rx(90), _t(0, -0.5, 0)]),
            recv("cone", "pantone_yellow", mesh("models/cone.obj", "Cone"), [_s(5.0), _t(0, 0, 0)])]})
    objects.append(light("large_light1", 9.0, [(0, 0.560784, 0), (3.5, 0.560784, 0), (5, 0.560784, 110), (7, 0.560784, 150)], [_rx(90), _t(0, 48, 0)]))
    # heroes on the stage
    heroes = [("buddha", "models/buddha.obj", "buddha", "gold_metallic_paint", 14.0), ("lucy", "models/lucy.obj", "lucy", "silver_paint", 16.0),
              ("ajax", "models/ajax.obj", "Ajax", "brass", 13.0), ("u_logo", "models/u_logo.obj", "U_Logo", "u_logo_plastic", 9.0),
              ("teapot2", "models/teapot.obj", "Teapot", "pantone_yellow_plastic", 6.0)]
    for i, (name, f, m, mat, size) in enumerate(heroes):
        x, z = ring(i, 5, 18.0)
        objects.append(recv(name, mat, mesh(f, m), [_s(size), _ry(72.0 * i), _t(x, size * 0.35, z)]))
    objects.append({"type": "group", "name": "red_teapot", "transform": [_t(-8, 0, -12)], "objects": [
        recv("base", "rough_silver_metal", mesh("models/teapot2.obj", "Base"), [_s(5.0), _t(0, 1.8, 0)]),
        recv("top", "rough_silver_metal", mesh("models/teapot2.obj", "Top"), [_s(3.0), _t(0, 4.2, 0)])]})
    objects.append(light("buddha_highlight", 3.0, [(0, 0.560784, 0), (4, 0.560784, 0), (5.5, 0.560784, 120)], [_rx(90), _t(0, 26, 18)]))
    objects.append(recv("dragon", "blue_acrylic", mesh("models/dragon.obj", "dragon"), keyframes=spline(
        [(_s(11.0), _ry(30.0 * k), _t(8 - 4 * k, 4 + 2 * k * (3 - k) / 2.0, -6 + k)) for k in range(4)], 11.5, 14.0)))
    for i in range(4):
        x, z = ring(i, 4, 30.0)
        tree = "models/kenny_nl/Tree_0%d.obj" % (1 + (i % 2 == 1))
        mats = ("glass", "red_wall") if i % 3 == 0 else ("green_plastic", "blue_acrylic")
        objects.append({"type": "group", "name": "kenny_nl_pine%d" % (i + 1), "transform": [_s(7.0), _ry(50.0 * i), _t(x * 0.9, 0, z * 0.9 + 4)], "objects": [
            recv("leaves", mats[0], mesh(tree, "Leaves"), [_s(1.0), _t(0, 1.1, 0)]),
            recv("trunk", mats[1], mesh(tree, "Trunk"), [_s([0.25, 0.8, 0.25]), _t(0, 0.3, 0)])]})
    objects.append(light("lucy_logo_section_light", 4.0, [(6, 0.560784, 0), (9, 0.54509, 140)], [_rx(90), _t(17, 30, 6)]))
    objects.append(light("left_backlight", 4.0, [(6, 0.560784, 0), (9, 0.54509, 140)], [_rx(60), _t(-25, 28, 20)]))
    objects.append(recv("rust_logo", "oxidized_steel", mesh("models/rust_logo.obj", "rust_logo"), keyframes=spline(
        [(_s(6.0 + 0.2 * k), _rx(15.0 * k), _ry(33.0 * k), _t(-12 + 2 * k, 6 + 1.5 * math.sin(k), -4 + 0.5 * k)) for k in range(12)], 12.0, 25.0)))
    objects.append({"type": "group", "name": "ajax_light_cone1", "transform": [_t(10.6, 30, -14.6)], "objects": [
        light("cone_light", 2.0, [(5, 0.560784, 0), (7, 0.760784, 350)], [_rx(90), _t(0, -0.5, 0)]),
        recv("cone", "pantone_yellow", mesh("models/cone.obj", "Cone"), [_s(5.0), _t(0, 0, 0)])]})
    objects.append(light("ajax_upper_light", 3.0, [(5, 0.560784, 0), (7, 0.760784, 160)], [_rx(90), _t(-10.6, 34, -14.6)]))
    objects.append(recv("cow", "glass", mesh("models/cow.obj", "Cow"), keyframes=spline(
        [(_s(5.0), _ry(50.0 * k), _t(-20 + 6 * k, 2.5 + (k % 2), -18 + 2 * k)) for k in range(7)], 18.0, 24.5)))
    orn = [("leaves", "Leaves", "forest_green_matte"), ("trunk", "Trunk", "sienna_matte"), ("tinsel", "Tinsel", "silver_metal"),
           ("sphere1", "Sphere1", "pantone_yellow"), ("sphere2", "Sphere2", "silver_metal"), ("sphere3", "Sphere3", "u_logo_plastic_shiny"),
           ("sphere4", "Sphere4", "glass"), ("sphere5", "Sphere5", "silver_paint"), ("bunny", "Bunny", "gold_metallic_paint"),
           ("suzanne", "Suzanne", "pantone_yellow_plastic")]
    for g in range(2):
        kids = []
        for j, (name, model, mat) in enumerate(orn):
            if j < 3:
                xf = [_s([1.0, 1.2, 1.0] if j != 1 else [0.25, 0.8, 0.25]), _t(0, 1.0 if j != 1 else 0.3, 0)]
            else:
                ox, oz = ring(j - 3, 7, 0.55)
                xf = [_s(0.22), _t(ox, 0.5 + 0.12 * j, oz)]
            kids.append(recv(name, mat if not (g == 1 and name == "suzanne") else "silver_paint", mesh("models/kenny_nl/tree_1_ornamented.obj", model), xf))
        objects.append({"type": "group", "name": "kenny_pine_ornaments" + ("" if g == 0 else "2"), "transform": [_s(9.0), _t(-26 + 52 * g, 0, -8)], "objects": kids})
    return {"film": dict(_film(width, height, samples), frames=int(frames), start_frame=0, end_frame=int(frames) - 1, scene_time=scene_time),
            "camera": {"fov": 40, "keyframes": spline(cam_pts, 0.0, scene_time)},
            "integrator": {"type": "pathtracer", "min_depth": 5, "max_depth": 10}, "materials": materials, "objects": objects}


def write_tr15_like_assets(directory, film=(1920, 1080, 512), frames=600, scene_time=25.0, detail=1.0):
    """tr15_like.json + its 13 OBJ files + 5 MERL tables under `directory`. detail scales every mesh grid (1.0 = about 3.1 M
    triangles in total; tests use ~0.03). Returns (scene path, triangle count)."""
    tris = 0
    seed = 100
    for path, models in sorted(_TR15_MODELS.items()):
        objs = []
        for name, gu, gv in models:
            seed += 1
            objs.append((name,) + knot_mesh(max(4, int(round(gu * detail))), max(3, int(round(gv * detail))), seed=seed, p=2 + seed % 2, q=3 + 2 * (seed % 3 == 0), extent=1.0))
        tris += write_obj(os.path.join(directory, path), objs)
    for name, (f, kd, ks, alpha) in _TR15_BRDFS.items():
        write_merl_binary(os.path.join(directory, f), kd=kd, ks=ks, alpha=alpha)
    return write_scene(tr15_like(*film, frames=frames, scene_time=scene_time), os.path.join(directory, "tr15_like.json")), tris


def write_png(path, pixels, width, height, channels):
    """8-bit PNG writer (channels: 1 grey, 2 grey+alpha, 3 RGB, 4 RGBA; non-interlaced, filter 0): the image textures of
    textured_box() and of the tests. `pixels` = bytes, row-major, top row first."""
    import struct
    import zlib
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[channels]
    stride = width * channels
    raw = b"".join(b"\x00" + bytes(pixels[y * stride:(y + 1) * stride]) for y in range(height))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, 8, ctype, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
    return path


def textured_box(width=160, height=120, samples=16, scene_time=0.0, shutter_size=0.0):
    """cornell_box with every kind of texture use of the reference (scene.rs:317-394, texture/*.rs): an RGB image as the diffuse colour of
    the back wall and floor, a grey image as the roughness of the ceiling (sample_f32 reads the red channel), an RGBA image as
    the gloss colour of the tall cube (a mesh: interpolated uv), an animated_image (two keyed frames, lerped at ray.time) on the
    left wall and a movie (three numbered frames at 2 frames per second) on the right wall. write_textured_box() writes the images."""
    d = cornell_box(width, height, samples)
    if scene_time > 0.0:
        d["film"].update({"frames": 2, "start_frame": 0, "end_frame": 1, "scene_time": scene_time})
        d["camera"]["shutter_size"] = shutter_size
    d["textures"] = [
        {"name": "checker", "type": "image", "file": "textures/checker.png"},
        {"name": "rough_map", "type": "image", "file": "textures/rough.png"},
        {"name": "gloss_map", "type": "image", "file": "textures/gloss.png"},
        {"name": "blink", "type": "animated_image", "keyframes": [{"file": "textures/blink_a.png", "time": 0.0},
                                                                   {"file": "textures/blink_b.png", "time": max(scene_time, 1.0)}]},
        {"name": "film_strip", "type": "movie", "file_prefix": "textures/strip_", "file_suffix": ".png", "frames": 3, "framerate": 2},
    ]
    d["materials"] += [
        {"type": "matte", "name": "checker_wall", "diffuse": "checker", "roughness": 1.0},
        {"type": "matte", "name": "rough_ceiling", "diffuse": [0.74, 0.74, 0.73], "roughness": "rough_map"},
        {"type": "plastic", "name": "gloss_plastic", "diffuse": [0.6, 0.6, 0.7], "gloss": "gloss_map", "roughness": 0.3},
        {"type": "matte", "name": "blink_wall", "diffuse": "blink", "roughness": 0.0},
        {"type": "matte", "name": "strip_wall", "diffuse": "film_strip", "roughness": 1.0},
    ]
    walls = d["objects"][0]["objects"]
    for w, m in zip(walls, ["checker_wall", "blink_wall", "strip_wall", "rough_ceiling", "checker_wall"]):
        w["material"] = m
    d["objects"][2]["material"] = "gloss_plastic"
    return d


def write_textured_box(directory, **kw):
    import random
    os.makedirs(os.path.join(directory, "models"), exist_ok=True)
    os.makedirs(os.path.join(directory, "textures"), exist_ok=True)
    with open(os.path.join(directory, "models", "cube.obj"), "w") as f:
        f.write(cube_obj())
    rnd = random.Random(5)
    t = os.path.join(directory, "textures")
    w, h = 16, 12
    write_png(os.path.join(t, "checker.png"), bytes(v for y in range(h) for x in range(w)
                                                    for v in ((200, 60, 40) if (x // 2 + y // 2) % 2 else (40 + 10 * x, 180, 220 - 12 * y))), w, h, 3)
    write_png(os.path.join(t, "rough.png"), bytes((x * 13 + y * 7) % 256 if (x + y) % 5 else 0 for y in range(8) for x in range(8)), 8, 8, 1)   # zeros: Lambertian texels
    write_png(os.path.join(t, "gloss.png"), bytes(v for y in range(6) for x in range(6)
                                                  for v in ((0, 0, 0, 255) if (x + y) % 4 == 0 else (rnd.randrange(256), rnd.randrange(256), rnd.randrange(256), 128))), 6, 6, 4)
    write_png(os.path.join(t, "blink_a.png"), bytes(v for y in range(4) for x in range(4) for v in (250 - 40 * x, 30 + 50 * y, 60)), 4, 4, 3)
    write_png(os.path.join(t, "blink_b.png"), bytes(v for y in range(5) for x in range(3) for v in (20, 240 - 30 * y, 90 + 60 * x)), 3, 5, 3)   # frames may differ in size
    for k in range(3):
        write_png(os.path.join(t, "strip_%05d.png" % k), bytes(v for y in range(4) for x in range(8) for v in ((80 * k + 20 * x) % 256, 200 - 40 * k, (30 * y + 70 * k) % 256, 255)), 8, 4, 4)
    return write_scene(textured_box(**kw), os.path.join(directory, "textured_box.json"))


# ---- AnimatedMesh (geometry/animated_mesh.rs): no scene of the reference uses one (its loader cannot construct it); this is the test scene
def flag_obj(grid, phase, amplitude=1.5, size=12.0):
    """a (grid x grid)-quad sheet in the xy plane, z = amplitude * sin(2 pi x / size + phase) * (x - x0) / size: the same topology for every
    phase, analytic normals, texcoords = the sheet's parameters"""
    import math
    lines = ["o Flag"]
    pos, nrm, tex = [], [], []
    for j in range(grid + 1):
        for i in range(grid + 1):
            u, v = i / grid, j / grid
            x, y = (u - 0.5) * size, (v - 0.5) * size
            k = 2.0 * math.pi / size
            z = amplitude * math.sin(k * x + phase) * u
            dz_dx = amplitude * (k * math.cos(k * x + phase) * u + math.sin(k * x + phase) / size)
            n = (-dz_dx, 0.0, 1.0)
            ln = math.sqrt(n[0] ** 2 + 1.0)
            pos.append((x, y, z)); nrm.append((n[0] / ln, 0.0, 1.0 / ln)); tex.append((u, v))
    lines += ["v %.6f %.6f %.6f" % p_ for p_ in pos]
    lines += ["vt %.6f %.6f" % t_ for t_ in tex]
    lines += ["vn %.6f %.6f %.6f" % n_ for n_ in nrm]
    for j in range(grid):
        for i in range(grid):
            a_ = j * (grid + 1) + i + 1
            b_, c_, d_ = a_ + 1, a_ + grid + 2, a_ + grid + 1
            lines.append("f " + " ".join("%d/%d/%d" % (q, q, q) for q in (a_, b_, c_, d_)))
    return "\n".join(lines) + "\n"


def waving_flag(width=160, height=120, samples=16, frames=8, scene_time=2.0, shutter_size=0.5, n_keys=4, times=None):
    """Cornell walls, the disk light, and a sheet that waves: an "animated_mesh" (animated_mesh.rs:14-27) of n_keys keyframes spread over
    the scene's time, under a static transform; a mirror ball so that paths meet the sheet from both sides."""
    d = cornell_box(width, height, samples)
    d["film"].update({"frames": frames, "start_frame": 0, "end_frame": frames - 1, "scene_time": scene_time})
    d["camera"] = {"fov": 30, "shutter_size": shutter_size, "transform": [_t(0, 12, -60)]}
    d["materials"] += [{"type": "metal", "name": "metal", "refractive_index": [0.155265, 0.116723, 0.138381],
                        "absorption_coefficient": [4.82835, 3.12225, 2.14696], "roughness": 0.2}]
    if times is None:
        times = [scene_time * k / (n_keys - 1) for k in range(n_keys)]
    walls = d["objects"][0]
    light = [o for o in d["objects"] if o.get("type") == "emitter"][0]
    flag = {"name": "flag", "type": "receiver", "material": "white_plastic",
            "geometry": {"type": "animated_mesh", "model": "Flag",
                         "keyframes": [{"file": "models/flag_%d.obj" % k, "time": times[k]} for k in range(n_keys)]},
            "transform": [_ry(25), _t(-1, 11, 4)]}
    ball = {"name": "ball", "type": "receiver", "material": "metal", "geometry": {"type": "sphere", "radius": 3.0}, "transform": [_t(7, 3, -4)]}
    d["objects"] = [walls, light, flag, ball]
    return d


def write_waving_flag(directory, grid=12, n_keys=4, **kw):
    import math
    os.makedirs(os.path.join(directory, "models"), exist_ok=True)
    with open(os.path.join(directory, "models", "cube.obj"), "w") as f:
        f.write(cube_obj())
    for k in range(n_keys):
        with open(os.path.join(directory, "models", "flag_%d.obj" % k), "w") as f:
            f.write(flag_obj(grid, 2.0 * math.pi * k / n_keys))
    return write_scene(waving_flag(n_keys=n_keys, **kw), os.path.join(directory, "waving_flag.json"))


def write_scene(scene, path):
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        json.dump(scene, f, indent=1)
    return path


def write_assets(directory, cornell=(800, 600, 4), small=(800, 600, 16)):
    """Write both scenes and the cube model under `directory`; returns (cornell_path, smallpt_path)."""
    os.makedirs(os.path.join(directory, "models"), exist_ok=True)
    with open(os.path.join(directory, "models", "cube.obj"), "w") as f:
        f.write(cube_obj())
    a = write_scene(cornell_box(*cornell), os.path.join(directory, "cornell_box.json"))
    b = write_scene(smallpt(*small), os.path.join(directory, "smallpt.json"))
    return a, b


if __name__ == "__main__":
    import sys
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes")
    print(write_assets(out))
