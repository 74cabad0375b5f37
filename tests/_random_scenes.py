"""Random static scenes in the reference's JSON schema for parity sweeps: every material kind, every geometry, point and area
lights, nested groups, all transform ops, both reconstruction filters, assorted integrator depths. Deterministic per seed."""
import json
import os

import numpy as np

from tray_rust_amd import scenes as S


def _color(rng, lo=0.05, hi=0.95):
    return [float(x) for x in rng.uniform(lo, hi, 3)]


def _material(rng, i, kinds):
    k = kinds[i % len(kinds)] if i < len(kinds) else rng.choice(kinds)
    name = f"m{i}_{k}"
    if k == "matte":
        return {"type": "matte", "name": name, "diffuse": _color(rng), "roughness": float(rng.choice([0.0, 1.0, 20.0, 60.0]))}
    if k == "plastic":
        return {"type": "plastic", "name": name, "diffuse": _color(rng), "gloss": _color(rng, 0.2, 0.9), "roughness": float(rng.uniform(0.05, 0.8))}
    if k == "metal":
        return {"type": "metal", "name": name, "refractive_index": _color(rng, 0.1, 2.0), "absorption_coefficient": _color(rng, 1.5, 5.0),
                "roughness": float(rng.uniform(0.05, 0.6))}
    if k == "specular_metal":
        return {"type": "specular_metal", "name": name, "refractive_index": _color(rng, 0.1, 2.0), "absorption_coefficient": _color(rng, 1.5, 5.0)}
    if k == "glass":
        return {"type": "glass", "name": name, "reflect": _color(rng, 0.7, 1.0), "transmit": _color(rng, 0.7, 1.0), "eta": float(rng.uniform(1.2, 1.8))}
    if k == "rough_glass":
        return {"type": "rough_glass", "name": name, "reflect": _color(rng, 0.7, 1.0), "transmit": _color(rng, 0.7, 1.0), "eta": float(rng.uniform(1.2, 1.8)),
                "roughness": float(rng.uniform(0.1, 0.6))}
    return {"type": "merl", "name": name, "file": "brdfs/blue-acrylic.binary"}


def _transform(rng, pos, scale=1.0):
    ops = []
    if rng.uniform() < 0.7:
        ops.append(S._s(float(scale * rng.uniform(0.6, 1.6))) if rng.uniform() < 0.5 else S._s([float(scale * x) for x in rng.uniform(0.5, 1.8, 3)]))
    for _ in range(int(rng.integers(0, 3))):
        kind = rng.choice(["rotate_x", "rotate_y", "rotate_z", "rotate"])
        if kind == "rotate":
            ops.append({"type": "rotate", "rotation": float(rng.uniform(-180, 180)), "axis": [float(x) for x in rng.normal(size=3)]})
        else:
            ops.append({"type": str(kind), "rotation": float(rng.uniform(-180, 180))})
    ops.append(S._t(*[float(x) for x in pos]))
    return ops


def _geometry(rng):
    g = rng.choice(["sphere", "disk", "rectangle", "plane", "cube", "knot"])
    if g == "sphere":
        return {"type": "sphere", "radius": float(rng.uniform(1.0, 3.5))}
    if g == "disk":
        r = float(rng.uniform(1.5, 4.0))
        return {"type": "disk", "radius": r, "inner_radius": float(rng.choice([0.0, 0.4 * r]))}
    if g == "rectangle":
        return {"type": "rectangle", "width": float(rng.uniform(2, 7)), "height": float(rng.uniform(2, 7))}
    if g == "plane":
        return {"type": "plane"}
    if g == "cube":
        return {"type": "mesh", "file": "models/cube.obj", "model": "Cube"}
    return {"type": "mesh", "file": "models/dragon.obj", "model": "dragon"}


def random_scene(seed, width=64, height=48, samples=8):
    rng = np.random.default_rng(seed)
    kinds = ["matte", "plastic", "metal", "specular_metal", "glass", "rough_glass", "merl"]
    rng.shuffle(kinds)
    d = S.cornell_box(width, height, samples)
    mats = [m for m in d["materials"]]
    n_extra = int(rng.integers(3, 8))
    mats += [_material(rng, i, kinds) for i in range(n_extra)]
    d["materials"] = mats
    walls = d["objects"][0]
    objects = [walls]
    if rng.uniform() < 0.3:   # re-dress the walls
        for wobj in walls["objects"]:
            wobj["material"] = str(rng.choice([m["name"] for m in mats]))
    # lights: at least one; point lights and area lights of the three allowed geometries
    for li in range(int(rng.integers(1, 4))):
        pos = rng.uniform([-10, 6, -12], [10, 22, 14])
        strength = float(rng.uniform(15, 60))
        if rng.uniform() < 0.35:
            objects.append({"name": f"point{li}", "type": "emitter", "emitter": "point", "emission": _color(rng, 0.5, 1.0) + [strength * 8],
                            "transform": [S._t(*[float(x) for x in pos])]})
        else:
            geom = {"sphere": {"type": "sphere", "radius": float(rng.uniform(0.8, 2.0))},
                    "disk": {"type": "disk", "radius": float(rng.uniform(1.5, 3.5)), "inner_radius": float(rng.choice([0.0, 0.7]))},
                    "rectangle": {"type": "rectangle", "width": float(rng.uniform(2, 6)), "height": float(rng.uniform(2, 6))}}[str(rng.choice(["sphere", "disk", "rectangle"]))]
            objects.append({"name": f"area{li}", "type": "emitter", "emitter": "area", "material": "white_wall", "emission": _color(rng, 0.5, 1.0) + [strength],
                            "geometry": geom, "transform": [S._rx(float(rng.choice([90.0, 60.0, 120.0])))] + _transform(rng, pos)[-1:]})
    # receivers, some inside a (possibly nested) group
    members = []
    for oi in range(int(rng.integers(2, 7))):
        g = _geometry(rng)
        scale = 8.0 if g.get("model") == "dragon" else (3.0 if g["type"] in ("plane", "mesh") else 1.0)
        obj = {"name": f"obj{oi}", "type": "receiver", "material": str(rng.choice([m["name"] for m in mats])), "geometry": g,
               "transform": _transform(rng, rng.uniform([-11, 1, -12], [11, 18, 14]), scale)}
        (members if rng.uniform() < 0.3 else objects).append(obj)
    if members:
        inner = {"type": "group", "name": "inner", "transform": _transform(rng, rng.uniform(-2, 2, 3)), "objects": members[1:]} if len(members) > 1 else None
        objects.append({"type": "group", "name": "outer", "transform": _transform(rng, rng.uniform(-2, 2, 3)),
                        "objects": members[:1] + ([inner] if inner else [])})
    d["objects"] = objects
    lo = int(rng.integers(0, 5))
    d["integrator"] = {"type": "pathtracer", "min_depth": lo, "max_depth": int(lo + rng.integers(0, 6))}
    if rng.uniform() < 0.5:
        d["film"]["filter"] = {"type": "gaussian", "width": float(rng.choice([1.0, 1.5, 2.0])), "height": 2.0, "alpha": float(rng.uniform(0.5, 3.0))}
    else:
        d["film"]["filter"] = {"type": "mitchell_netravali", "width": float(rng.choice([1.0, 1.5, 2.0])), "height": 2.0,
                               "b": float(rng.uniform(0, 1)), "c": float(rng.uniform(0, 1))}
    if rng.uniform() < 0.5:
        d["camera"] = {"fov": float(rng.uniform(25, 70)), "position": [float(x) for x in rng.uniform([-8, 6, -58], [8, 18, -40])],
                       "target": [float(x) for x in rng.uniform([-3, 8, -2], [3, 14, 4])], "up": [0, 1, 0]}
    else:
        d["camera"] = {"fov": float(rng.uniform(25, 45)), "transform": [S._ry(float(rng.uniform(-6, 6))), S._t(float(rng.uniform(-3, 3)), 12, -60)]}
    return d


def write_random_scene(directory, seed, **kw):
    """<dir>/random_<seed>.json plus the assets any random scene may name (cube, a small knot mesh, one MERL table)"""
    os.makedirs(os.path.join(directory, "models"), exist_ok=True)
    cube = os.path.join(directory, "models", "cube.obj")
    if not os.path.exists(cube):
        open(cube, "w").write(S.cube_obj())
        S.write_dragon_obj(os.path.join(directory, "models", "dragon.obj"), grid=12, extent=0.6)
        S.write_merl_binary(os.path.join(directory, "brdfs", "blue-acrylic.binary"))
    path = os.path.join(directory, f"random_{seed}.json")
    json.dump(random_scene(seed, **kw), open(path, "w"))
    return path
