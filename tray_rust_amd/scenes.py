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
