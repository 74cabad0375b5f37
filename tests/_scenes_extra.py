"""Small closed-form scenes shared by the CPU (oracle vs formula) and GPU (kernels vs oracle) tests."""
from tray_rust_amd import scenes


def sliding_point_light():
    """Lambertian floor, camera straight above the origin, a point light with keyed emission sliding along x under an open shutter."""
    d = scenes.cornell_box(8, 8, 4)
    d["film"].update({"frames": 1, "scene_time": 1.0})
    d["integrator"] = {"type": "pathtracer", "min_depth": 0, "max_depth": 0}
    d["materials"] = [{"type": "matte", "name": "m", "diffuse": [0.5, 0.5, 0.5], "roughness": 0.0}]
    tr = lambda x: {"transform": [{"type": "translate", "translation": [x, 4.0, 0.0]}]}
    d["objects"] = [
        {"name": "floor", "type": "receiver", "material": "m", "geometry": {"type": "rectangle", "width": 400, "height": 400},
         "transform": [{"type": "rotate_x", "rotation": -90}]},
        {"name": "spark", "type": "emitter", "emitter": "point",
         "emission": [{"time": 0.0, "color": [1, 1, 1, 10]}, {"time": 1.0, "color": [1, 1, 1, 30]}],   # stays below the per-sample clamp (Q3)
         "keyframes": {"control_points": [tr(-6.0), tr(6.0)], "knots": [0, 0, 1, 1], "degree": 1}},
    ]
    d["camera"] = {"fov": 1.0, "shutter_size": 1.0, "transform": [{"type": "rotate_x", "rotation": 90}, {"type": "translate", "translation": [0, 5, 0]}]}
    return d


def crossing_emitter():
    """A unit sphere emitter crossing the almost parallel camera rays of the film at constant speed under an open shutter."""
    d = scenes.cornell_box(8, 8, 4)
    d["film"].update({"frames": 1, "scene_time": 1.0})
    d["integrator"] = {"type": "pathtracer", "min_depth": 0, "max_depth": 0}
    d["materials"] = [{"type": "matte", "name": "m", "diffuse": [0.0, 0.0, 0.0], "roughness": 0.0}]
    tr = lambda x: {"transform": [{"type": "translate", "translation": [x, 2.0, 0.0]}]}
    d["objects"] = [{"name": "ball", "type": "emitter", "emitter": "area", "material": "m", "emission": [1, 1, 1, 0.75],
                     "geometry": {"type": "sphere", "radius": 1.0},
                     "keyframes": {"control_points": [tr(-3.0), tr(3.0)], "knots": [0, 0, 1, 1], "degree": 1}}]
    d["camera"] = {"fov": 0.5, "shutter_size": 1.0, "transform": [{"type": "rotate_x", "rotation": 90}, {"type": "translate", "translation": [0, 50, 0]}]}
    return d
