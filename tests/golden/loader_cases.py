"""Inputs of the asset-loader fixtures: small OBJ texts and .map arrays that exercise the quirks of the reference's
loaders (src/objectunion.cpp:144-443, src/valuemap.cpp:54-100).  tests/golden/make_golden.py feeds them to the
reference itself and stores what it produced in loader_obj.npz / loader_map.npz; tests/test_host_loader.py feeds
them to this repository's loader and compares bit for bit.  Only inputs the reference handles without undefined
behaviour are here (a face line with a trailing space, for one, makes it store an uninitialised pointer)."""
import numpy as np

_GEOMETRY = """# a few vertices, texture coordinates and normals
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0.5 0.5 1e0
v -2.5e-1 +0.75 .5
v 3.25 -1.5 2 1.0
vt 0 0
vt 1 0
vt 1 1
vt 0 1 0.5
vn 0 0 1
vn 0 1 0
vn 1 0 0
vn 0.6 0 0.8
"""

_FACES = """f 1/1/1 2/2/2 3/3/3
f 1/1/1 3/3/3 4/4/1 5/1/2
f 1//1 2//2 5//3
f 1/1 2/2 6/3
f 1 2 6
f 1 2 3 4 5
f 2/2/2 3/3 4
f 7/1/4 5/2/4 6/3/4
 f 1 2 3
f\t1 2 3
f 1  2 3
g ignored
f 4/4/4 3/3/3 2/2/2 1/1/1
"""

OBJ_TEXTS = {
    "formats": _GEOMETRY + _FACES,
    "crlf": (_GEOMETRY + _FACES).replace("\n", "\r\n"),
    "no_final_newline": (_GEOMETRY + _FACES).rstrip("\n"),
    "vertices_only": _GEOMETRY,
    "double_space_in_vertex": "v  1 0 0\nv 0 1 0\nv 0 0 1\nf 1 2 3\n",
}

_ORIENTATION = {"X": 20.0, "Y": 35.0, "Z": -10.0}
# (name, OBJ text key, parameters of the ObjectUnion in scene.json)
OBJ_CASES = [
    ("flat", "formats", dict(enable_smooth_shading=False)),
    ("smooth", "formats", dict(enable_smooth_shading=True)),
    ("default_shading", "formats", dict()),
    ("crlf", "crlf", dict(enable_smooth_shading=True)),
    ("no_final_newline", "no_final_newline", dict(enable_smooth_shading=True)),
    ("center_only", "formats", dict(center=[0.25, -1.0, 2.0])),
    ("center_size", "formats", dict(center=[0.25, -1.0, 2.0], size=0.75, enable_smooth_shading=True)),
    ("center_size_orientation", "formats", dict(center=[0.0, 0.62, -0.8], size=0.6, orientation=_ORIENTATION, enable_smooth_shading=True)),
    ("center_orientation", "formats", dict(center=[1.0, 2.0, 3.0], orientation=_ORIENTATION)),
    ("size_without_center", "formats", dict(size=3.0, orientation=_ORIENTATION)),
    ("vertices_only", "vertices_only", dict(center=[0.0, 0.0, 0.0], size=1.0)),
    ("double_space_in_vertex", "double_space_in_vertex", dict()),
    ("missing_file", None, dict()),
]


def map_arrays():
    rng = np.random.default_rng(17)
    return {
        "rgb": rng.random((5, 7, 3)),
        "grey": rng.random((4, 3, 1)),
        "single_texel": np.array([[[0.2, 0.5, 0.9]]]),
        "dark_values": np.array([[[0.0, 0.002, 0.0031308], [0.04045, 0.05, 1.0]]]),  # both branches of the sRGB curve
    }


# (name, array key or raw float64 stream, channels asked for, use_gamma_correction)
MAP_CASES = [
    ("rgb_gamma", "rgb", 3, True),
    ("rgb_linear", "rgb", 3, False),
    ("grey", "grey", 1, True),
    ("single_texel", "single_texel", 3, True),
    ("dark_values", "dark_values", 3, True),
    ("size_mismatch", np.array([4.0, 4.0, 3.0, 0.5, 0.5]), 3, False),
    ("negative_header", np.array([-1.0, 1.0, 3.0, 0.5, 0.5, 0.5]), 3, False),
    ("missing_file", None, 3, False),
]


def write_inputs(directory):
    """Writes every input file under `directory`; returns {obj text key: path}, {map case name: path}."""
    import os
    os.makedirs(directory, exist_ok=True)
    obj_paths, map_paths = {}, {}
    for key, text in OBJ_TEXTS.items():
        obj_paths[key] = os.path.join(directory, key + ".obj")
        with open(obj_paths[key], "wb") as f:
            f.write(text.encode())
    arrays = map_arrays()
    for name, source, _, _ in MAP_CASES:
        path = os.path.join(directory, name + ".map")
        map_paths[name] = path
        if source is None:
            continue
        if isinstance(source, str):
            a = arrays[source]
            stream = np.concatenate([[a.shape[1], a.shape[0], a.shape[2]], a.ravel()]).astype(np.float64)
        else:
            stream = source
        stream.tofile(path)
    return obj_paths, map_paths
