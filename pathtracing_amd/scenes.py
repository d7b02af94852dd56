"""Scene-directory generators for tests and benchmarks.

Each builder writes a directory in the reference's scene-directory layout (settings.json,
scene.json, denoising.json; meshes and textures beside them) so that it can be given to
`pathtracer <scene_directory>` unchanged.  The shipped scenes of the reference cannot travel to
the GPU box, and its living-room assets are not in its repository at all, so the workloads named
by the benchmark configuration are generated here:

* ``cornell_example``   the Cornell box of the reference's scenes/example (same parameters)
* ``material_zoo``      every material / object / light / camera feature in one small box
* ``mesh_box``          Cornell walls + a procedurally generated OBJ mesh (up to millions of triangles)
* ``cloud_box``         the Cornell box with a scattering-medium volume and coloured glass
* ``living_room``       stand-in for the reference's living-room scene: 23 objects, OBJ meshes with file
                        textures, a textured floor and wall, a small sphere light
"""
import json
import os

import numpy as np

DEFAULT_DENOISING = {
    "pipeline": [
        {"mode": "median", "parameters": {"kernel_size": 3, "threshold": 0.5}},
        {"mode": "atrous", "parameters": {"iterations": 4, "sigma_rt": 2, "sigma_x": 2, "sigma_n": 2}},
    ]
}


def default_settings(**overrides):
    s = {
        "WIDTH": 1000, "HEIGHT": 1000, "samples_per_pixel": 100, "samples_per_iteration": 16,
        "max_recursion_depth": 100, "min_recursion_steps": 3, "number_of_threads": 1,
        "use_gamma_correction": True, "bvh_leaf_size": 4, "bvh_n_axis_splits": 16,
        "enable_next_event_estimation": True, "enable_anti_aliasing": True,
    }
    s.update(overrides)
    return s


def _map3(name, rgb):
    return {"name": name, "type": "ValueMap3D", "parameters": {"data": list(rgb)}}


def _map1(name, value):
    return {"name": name, "type": "ValueMap1D", "parameters": {"data": [value]}}


def _map_file(name, kind, path):
    return {"name": name, "type": kind, "parameters": {"file": path}}


def _medium(name, subtype, scattering, absorption, ior, emission=(0, 0, 0)):
    return {"name": name, "type": "Medium", "subtype": subtype,
            "parameters": {"scattering_albedo": list(scattering), "absorption_albedo": list(absorption),
                           "emission_coefficient": list(emission), "refractive_index": ior}}


def _material(name, subtype, **parameters):
    return {"name": name, "type": "Material", "subtype": subtype, "parameters": parameters}


def _rect(name, position, v1, v2, L1, L2, material):
    return {"name": name, "type": "Object", "subtype": "Rectangle",
            "parameters": {"position": position, "v1": v1, "v2": v2, "L1": L1, "L2": L2, "material": material}}


def _sphere(name, position, radius, material):
    return {"name": name, "type": "Object", "subtype": "Sphere",
            "parameters": {"material": material, "position": position, "radius": radius}}


def _camera(position, X, Y, Z, width, focal, **dof):
    cam = {"camera_position": position, "orientation": {"X": X, "Y": Y, "Z": Z}, "camera_width": width,
           "focal_length": focal}
    cam.update(dof)
    return cam


def _cornell_shell(wall_material="white_diffuse_material"):
    """The six walls and the ceiling light of the example box (2.2 x 2 x 4 units)."""
    return [
        _rect("floor", [0, 0, 0], [1, 0, 0], [0, 0, -1], 2.2, 4, wall_material),
        _rect("front_wall", [0, 1, -2], [1, 0, 0], [0, 1, 0], 2.2, 2, wall_material),
        _rect("left_wall", [-1.1, 1, 0], [0, 0, -1], [0, 1, 0], 4, 2, "red_diffuse_material"),
        _rect("right_wall", [1.1, 1, 0], [0, 0, 1], [0, 1, 0], 4, 2, "green_diffuse_material"),
        _rect("back_wall", [0, 1, 2], [1, 0, 0], [0, -1, 0], 2.2, 2, wall_material),
        _rect("roof", [0, 2, 0], [1, 0, 0], [0, 0, 1], 2.2, 4, wall_material),
    ]


_LIGHT = _rect("light_source", [0, 1.9999, -1], [1, 0, 0], [0, 0, 1], 0.4, 0.4, "light_source_material")

_BASE_MAPS = [_map3("red", [0.8, 0, 0]), _map3("green", [0, 0.6, 0]), _map3("blue", [0, 0, 0.8]),
              _map3("cyan", [0, 0.6, 0.8]), _map3("gold", [1.0, 0.84, 0.05]), _map3("white", [1, 1, 1]),
              _map1("gold_roughness", 0.2), _map1("glass_roughness", 0.15), _map1("light_strength", 40)]

_BASE_MEDIA = [_medium("air_medium", "BeersLawMedium", [0, 0, 0], [0, 0, 0], 1.0),
               _medium("glass_medium", "BeersLawMedium", [0, 0, 0], [0, 0, 0], 1.5),
               _medium("cyan_glass_medium", "BeersLawMedium", [0, 0, 0], [10, 2, 2], 1.5)]

_BASE_MATERIALS = [
    _material("red_diffuse_material", "Diffuse", albedo_map="red"),
    _material("green_diffuse_material", "Diffuse", albedo_map="green"),
    _material("blue_diffuse_material", "Diffuse", albedo_map="blue", surface_refractive_index=1.5),
    _material("cyan_glass_material", "Transparent", albedo_map="cyan", surface_refractive_index=1.5,
              internal_medium="cyan_glass_medium", external_medium="air_medium"),
    _material("gold_material", "MetallicMicrofacet", albedo_map="gold", surface_refractive_index=0.3,
              extinction_coefficient=2.9, roughness_map="gold_roughness"),
    _material("rough_glass_material", "TransparentMicrofacet", roughness_map="glass_roughness",
              internal_medium="glass_medium", external_medium="air_medium", surface_refractive_index=1.5),
    _material("mirror_material", "ReflectiveMicrofacet", albedo_map="white", surface_refractive_index=1.5),
    _material("white_diffuse_material", "Diffuse", albedo_map="white"),
    _material("light_source_material", "Diffuse", albedo_map="white", emission_color_map="white",
              light_intensity_map="light_strength", is_light_source=True),
]

_EXAMPLE_CAMERA = _camera([0, 0.9, 1.9], 90, 0, 0, 0.036, 0.036)


def cornell_example_dict():
    objects = _cornell_shell() + [
        _sphere("blue_ball", [-0.35, 0.5, -1.3], 0.5, "blue_diffuse_material"),
        _sphere("glass_ball", [-0.5, 0.1, -0.5], 0.1, "cyan_glass_material"),
        _sphere("rough_glass_ball", [0, 0.25, -0.4], 0.25, "rough_glass_material"),
        _sphere("mirror_ball", [0.4, 0.15, -0.5], 0.15, "mirror_material"),
        _sphere("gold_ball", [0.5, 0.25, -1], 0.25, "gold_material"),
        _LIGHT,
    ]
    return {"valuemaps": list(_BASE_MAPS), "media": list(_BASE_MEDIA), "materials": list(_BASE_MATERIALS),
            "objects": objects, "background_medium": "air_medium", "background_color": [0.0, 0.0, 0.0],
            "camera": dict(_EXAMPLE_CAMERA)}


def write_scene_directory(directory, scene, settings=None, denoising=None):
    os.makedirs(directory, exist_ok=True)
    with open(os.path.join(directory, "scene.json"), "w") as f:
        json.dump(scene, f, indent=1)
    with open(os.path.join(directory, "settings.json"), "w") as f:
        json.dump(settings or default_settings(), f, indent=1)
    with open(os.path.join(directory, "denoising.json"), "w") as f:
        json.dump(denoising or DEFAULT_DENOISING, f, indent=1)
    return directory


def cornell_example(directory, **settings):
    """The benchmark's first workload: the reference's example Cornell box (12 analytic objects)."""
    return write_scene_directory(directory, cornell_example_dict(), default_settings(**settings))


# ------------------------------------------------------------------------------------------
# textures and meshes
# ------------------------------------------------------------------------------------------
def cornell_example_all_deferred(directory, **settings):
    """The example box plus a mesh of two specks outside opposite corners of the room, so that the bounds of the mesh
    hold the whole scene: every ray enters them, and every vertex takes the deferred route (shading kernel -> ray queue
    -> traversal kernel -> completion kernel).  The image is the example's; the scene measures what that route costs
    when the walk itself is free."""
    os.makedirs(os.path.join(directory, "models"), exist_ok=True)
    with open(os.path.join(directory, "models", "specks.obj"), "w") as f:
        for x, y, z in ((-1.15, -0.05, -2.05), (1.15, 2.05, 2.05)):
            f.write(f"v {x} {y} {z}\nv {x + 1e-3} {y} {z}\nv {x} {y + 1e-3} {z}\n")
        f.write("f 1 2 3\nf 4 5 6\n")
    scene = cornell_example_dict()
    scene["objects"] = scene["objects"] + [{"name": "specks", "type": "Object", "subtype": "ObjectUnion",
                                            "parameters": {"file": "models/specks.obj", "material": "white_diffuse_material",
                                                           "enable_smooth_shading": False}}]
    return write_scene_directory(directory, scene, default_settings(**settings))


def write_map_file(path, data):
    """`.map`: float64 [width, height, dim, data...] with data row-major [h][w][dim]."""
    a = np.ascontiguousarray(data, dtype=np.float64)
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, d = a.shape
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    np.concatenate([np.array([w, h, d], dtype=np.float64), a.ravel()]).tofile(path)


def value_noise(width, height, channels, seed, cells=8):
    """Smooth tileable value noise in [0, 1], [height, width, channels]."""
    rng = np.random.default_rng(seed)
    grid = rng.random((cells, cells, channels))
    ys = np.arange(height) * cells / height
    xs = np.arange(width) * cells / width
    y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
    fy, fx = (ys - y0)[:, None, None], (xs - x0)[None, :, None]
    fy, fx = fy * fy * (3 - 2 * fy), fx * fx * (3 - 2 * fx)
    y1, x1 = (y0 + 1) % cells, (x0 + 1) % cells
    a, b = grid[y0][:, x0], grid[y0][:, x1]
    c, d = grid[y1][:, x0], grid[y1][:, x1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def checker(width, height, cells, low, high):
    y, x = np.mgrid[0:height, 0:width]
    mask = ((x * cells // width) + (y * cells // height)) % 2
    return np.where(mask, high, low).astype(np.float64)


def write_bumpy_sphere_obj(path, n_lon, n_lat, seed=7, bump=0.08, quads=True):
    """UV sphere with a sinusoidal + seeded-noise radial displacement, written in the OBJ subset
    the loader accepts (single spaces; `v`, `vt`, `vn`, `f a/b/c`).  Triangles: 2*n_lon*(n_lat-1).
    n_lon=1024, n_lat=513 gives exactly 1,048,576."""
    rng = np.random.default_rng(seed)
    theta = np.linspace(0.0, np.pi, n_lat + 1)           # rows of vertices, poles included
    phi = np.linspace(0.0, 2 * np.pi, n_lon + 1)         # seam duplicated for clean UVs
    T, P = np.meshgrid(theta, phi, indexing="ij")
    noise = rng.random((n_lat + 1, n_lon + 1)) * 0.25
    noise[:, -1] = noise[:, 0]
    r = 1.0 + bump * (np.sin(5 * T) * np.sin(4 * P) + noise * np.sin(T))
    x, y, z = r * np.sin(T) * np.cos(P), r * np.cos(T), r * np.sin(T) * np.sin(P)
    nx, ny, nz = np.sin(T) * np.cos(P), np.cos(T), np.sin(T) * np.sin(P)
    u, v = P / (2 * np.pi), 1.0 - T / np.pi
    cols = n_lon + 1
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        f.write("# generated bumpy sphere\n")
        f.write("".join("v %.9g %.9g %.9g\n" % t for t in zip(x.ravel(), y.ravel(), z.ravel())))
        f.write("".join("vt %.9g %.9g\n" % t for t in zip(u.ravel(), v.ravel())))
        f.write("".join("vn %.9g %.9g %.9g\n" % t for t in zip(nx.ravel(), ny.ravel(), nz.ravel())))
        i, j = np.meshgrid(np.arange(n_lat), np.arange(n_lon), indexing="ij")
        a = (i * cols + j + 1).ravel()
        b = (i * cols + j + 2).ravel()
        c = ((i + 1) * cols + j + 2).ravel()
        d = ((i + 1) * cols + j + 1).ravel()
        rows = i.ravel()

        def write_faces(fmt, select, *corners):  # one line per selected cell, corner indices repeated for v/vt/vn
            cols_ = [np.repeat(c[select][:, None], 3, axis=1) for c in corners]
            table = np.concatenate(cols_, axis=1).tolist()
            for lo in range(0, len(table), 262144):
                f.write("".join([fmt % tuple(r) for r in table[lo:lo + 262144]]))

        tri = "f %d/%d/%d %d/%d/%d %d/%d/%d\n"
        north, south = rows == 0, rows == n_lat - 1
        inner = ~(north | south)
        write_faces(tri, north, a, c, d)                       # north-pole fan: one triangle per cell
        if quads:
            write_faces("f %d/%d/%d %d/%d/%d %d/%d/%d %d/%d/%d\n", inner, a, b, c, d)
        else:
            ai, bi, ci, di = a[inner], b[inner], c[inner], d[inner]
            two = np.stack([np.stack([ai, bi, ci], 1), np.stack([ai, ci, di], 1)], 1).reshape(-1, 3)
            write_faces(tri, slice(None), two[:, 0], two[:, 1], two[:, 2])
        write_faces(tri, south, a, b, d)                       # south-pole fan
    return 2 * n_lon * (n_lat - 1)


def mesh_box(directory, n_lon=64, n_lat=33, material="gold", smooth=True, textured=False, seed=7, **settings):
    """Cornell walls + ceiling light + one OBJ mesh placed with center/size/orientation.
    `material`: 'gold' (MetallicMicrofacet a=0.2), 'rough_glass' (TransparentMicrofacet),
    'glass' (Transparent, IOR 1.5) or 'diffuse'."""
    os.makedirs(directory, exist_ok=True)
    obj_path = os.path.join(directory, "models", "bumpy_sphere.obj")
    n_tri = write_bumpy_sphere_obj(obj_path, n_lon, n_lat, seed=seed)
    maps = list(_BASE_MAPS)
    materials = list(_BASE_MATERIALS)
    if textured:
        tex = value_noise(256, 256, 3, seed=11)
        write_map_file(os.path.join(directory, "maps", "noise_albedo.map"), tex)
        write_map_file(os.path.join(directory, "maps", "checker_roughness.map"), checker(128, 128, 16, 0.05, 0.4))
        maps += [_map_file("noise_albedo", "ValueMap3D", "maps/noise_albedo.map"),
                 _map_file("checker_roughness", "ValueMap1D", "maps/checker_roughness.map")]
        materials += [_material("textured_material", "Glossy", albedo_map="noise_albedo",
                                surface_refractive_index=1.3, roughness_map="checker_roughness")]
    materials += [_material("glass_material", "Transparent", internal_medium="glass_medium",
                            external_medium="air_medium"),
                  _material("mesh_diffuse_material", "Diffuse", albedo_map="blue")]
    name = {"gold": "gold_material", "rough_glass": "rough_glass_material", "glass": "glass_material",
            "diffuse": "mesh_diffuse_material", "textured": "textured_material"}[material if not textured else "textured"]
    mesh = {"name": "mesh", "type": "Object", "subtype": "ObjectUnion",
            "parameters": {"file": "models/bumpy_sphere.obj", "material": name, "enable_smooth_shading": bool(smooth),
                           "center": [0.0, 0.62, -0.8], "size": 0.6, "orientation": {"X": 20, "Y": 35, "Z": 10}}}
    scene = {"valuemaps": maps, "media": list(_BASE_MEDIA), "materials": materials,
             "objects": _cornell_shell() + [mesh, _LIGHT], "background_medium": "air_medium",
             "background_color": [0.0, 0.0, 0.0], "camera": dict(_EXAMPLE_CAMERA)}
    write_scene_directory(directory, scene, default_settings(**settings))
    return n_tri


def write_thicket_obj(path, n_triangles=6000, seed=5):
    """A mesh no tessellator would produce: triangles of geometrically growing size nested around one point
    (a surface-area hierarchy peels them off one at a time and grows deep), tight clusters of tiny triangles,
    and long thin triangles that span the whole mesh.  Plain `v` / `f a b c` lines."""
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    centres = rng.uniform(-0.8, 0.8, size=(7, 3))
    with open(path, "w") as f:
        f.write("# generated thicket\n")
        faces = []
        for i in range(n_triangles):
            kind = i % 4
            if kind < 2:    # nested around the origin
                c = np.zeros(3)
                extent = 1e-8 * 1.35 ** (i // 4 % 60)
                d = rng.normal(size=(3, 3)) * extent
            elif kind == 2:  # a cluster of tiny triangles
                c = centres[i // 4 % 7] + rng.normal(size=3) * 1e-3
                d = rng.normal(size=(3, 3)) * 2e-4
            else:            # long and thin
                c = rng.uniform(-0.5, 0.5, size=3)
                axis = i // 4 % 3
                d = rng.normal(size=(3, 3)) * 2e-4
                d[:, axis] = np.array([-1.0, 1.0, 0.9]) * rng.uniform(0.5, 1.0)
            for k in range(3):
                f.write("v %.12g %.12g %.12g\n" % tuple(c + d[k]))
            faces.append("f %d %d %d\n" % (3 * i + 1, 3 * i + 2, 3 * i + 3))
        f.write("".join(faces))
    return n_triangles


def thicket_box(directory, material="gold", n_triangles=6000, **settings):
    """Cornell walls + ceiling light + the mesh of write_thicket_obj (flat shading, no texture coordinates)."""
    os.makedirs(directory, exist_ok=True)
    write_thicket_obj(os.path.join(directory, "models", "thicket.obj"), n_triangles)
    name = {"gold": "gold_material", "rough_glass": "rough_glass_material"}[material]
    mesh = {"name": "mesh", "type": "Object", "subtype": "ObjectUnion",
            "parameters": {"file": "models/thicket.obj", "material": name, "enable_smooth_shading": False,
                           "center": [0.0, 0.9, -0.8], "size": 0.8, "orientation": {"X": 10, "Y": 25, "Z": 5}}}
    scene = {"valuemaps": list(_BASE_MAPS), "media": list(_BASE_MEDIA), "materials": list(_BASE_MATERIALS),
             "objects": _cornell_shell() + [mesh, _LIGHT], "background_medium": "air_medium",
             "background_color": [0.0, 0.0, 0.0], "camera": dict(_EXAMPLE_CAMERA)}
    write_scene_directory(directory, scene, default_settings(**settings))
    return directory


def material_zoo(directory, dof=None, **settings):
    """One small box exercising every material, object, light and texture path:
    Plane floor, top-level Triangle, Glossy / Reflective / Transparent / all microfacet types,
    a transparency-mapped (cut-out) rectangle, file textures, a sphere light and a rectangle light,
    a non-black background, and optionally a thin-lens camera (dof='circle'|'square')."""
    os.makedirs(directory, exist_ok=True)
    write_map_file(os.path.join(directory, "maps", "zoo_albedo.map"), value_noise(64, 32, 3, seed=3))
    write_map_file(os.path.join(directory, "maps", "zoo_cutout.map"), checker(32, 32, 8, 0.0, 1.0))
    write_map_file(os.path.join(directory, "maps", "zoo_rough.map"), 0.05 + 0.3 * value_noise(32, 32, 1, seed=5)[:, :, 0])
    maps = list(_BASE_MAPS) + [
        _map_file("zoo_albedo", "ValueMap3D", "maps/zoo_albedo.map"),
        _map_file("zoo_cutout", "ValueMap1D", "maps/zoo_cutout.map"),
        _map_file("zoo_rough", "ValueMap1D", "maps/zoo_rough.map"),
        _map1("half", 0.5), _map1("sphere_light_strength", 25), _map3("warm", [0.9922, 0.9569, 0.8627]),
        _map1("wood_roughness", 0.3),
    ]
    materials = list(_BASE_MATERIALS) + [
        _material("glossy_material", "Glossy", albedo_map="zoo_albedo", surface_refractive_index=1.3,
                  roughness_map="wood_roughness"),
        _material("glossy_metal_material", "Glossy", albedo_map="gold", surface_refractive_index=1.8,
                  roughness_map="zoo_rough", is_dielectric=False),
        _material("perfect_mirror_material", "Reflective", albedo_map="cyan", is_dielectric=False),
        _material("dielectric_mirror_material", "Reflective"),
        _material("clear_glass_material", "Transparent", internal_medium="glass_medium", external_medium="air_medium"),
        _material("cutout_material", "Diffuse", albedo_map="green", transparency_map="zoo_cutout"),
        _material("veil_material", "ReflectiveMicrofacet", albedo_map="white", roughness_map="gold_roughness",
                  transparency_map="half"),
        _material("textured_diffuse_material", "Diffuse", albedo_map="zoo_albedo"),
        _material("sphere_light_material", "Diffuse", albedo_map="white", emission_color_map="warm",
                  light_intensity_map="sphere_light_strength", is_light_source=True),
    ]
    objects = [
        {"name": "ground", "type": "Object", "subtype": "Plane",
         "parameters": {"position": [0, 0, 0], "v1": [2, 0, 0], "v2": [0, 0, -3], "material": "textured_diffuse_material"}},
        _rect("front_wall", [0, 1, -2], [1, 0, 0], [0, 1, 0], 2.2, 2, "white_diffuse_material"),
        _rect("left_wall", [-1.1, 1, 0], [0, 0, -1], [0, 1, 0], 4, 2, "red_diffuse_material"),
        _rect("right_wall", [1.1, 1, 0], [0, 0, 1], [0, 1, 0], 4, 2, "glossy_material"),
        _rect("roof", [0, 2, 0], [1, 0, 0], [0, 0, 1], 2.2, 4, "white_diffuse_material"),
        _sphere("glossy_ball", [-0.6, 0.3, -1.0], 0.3, "glossy_material"),
        _sphere("glossy_metal_ball", [0.65, 0.2, -0.2], 0.2, "glossy_metal_material"),
        _sphere("mirror_ball", [0.55, 0.3, -1.2], 0.3, "perfect_mirror_material"),
        _sphere("dielectric_mirror_ball", [-0.75, 0.15, -0.1], 0.15, "dielectric_mirror_material"),
        _sphere("clear_glass_ball", [0.0, 0.25, -0.6], 0.25, "clear_glass_material"),
        _sphere("cyan_glass_ball", [-0.3, 0.12, 0.2], 0.12, "cyan_glass_material"),
        _sphere("rough_glass_ball", [0.3, 0.15, 0.1], 0.15, "rough_glass_material"),
        _sphere("gold_ball", [-0.1, 0.2, -1.5], 0.2, "gold_material"),
        _sphere("veil_ball", [0.2, 0.75, -1.1], 0.18, "veil_material"),
        _rect("cutout_screen", [-0.3, 0.6, -0.2], [1, 0, 0.3], [0, 1, 0], 0.6, 0.6, "cutout_material"),
        {"name": "lone_triangle", "type": "Object", "subtype": "Triangle",
         "parameters": {"p1": [-1.0, 0.05, -1.8], "p2": [-0.2, 0.05, -1.9], "p3": [-0.6, 0.9, -1.7],
                        "material": "gold_material"}},
        _sphere("sphere_light", [-0.6, 1.6, -0.6], 0.12, "sphere_light_material"),
        _rect("light_source", [0.4, 1.9999, -1], [1, 0, 0], [0, 0, 1], 0.4, 0.4, "light_source_material"),
    ]
    camera = _camera([0, 0.9, 1.9], 88, 3, 0, 0.036, 0.036)
    if dof:
        camera.update({"depth_of_field_mode": dof, "focus_distance": 2.6, "aperture_size": 0.08})
    scene = {"valuemaps": maps, "media": list(_BASE_MEDIA), "materials": materials, "objects": objects,
             "background_medium": "air_medium", "background_color": [0.05, 0.07, 0.1], "camera": camera}
    return write_scene_directory(directory, scene, default_settings(**settings))


def random_box(directory, seed, extras=False, **settings):
    """A box with random contents drawn from `seed`: 4-14 spheres, rectangles and triangles at random places,
    each with a material of a random type and random parameters (every material type can occur, with constant
    or no maps, sometimes a constant transparency), one or two lights (a rectangle, a sphere, or both), a random
    background colour, up to two small OBJ meshes, and sometimes a scattering medium as the background medium or inside a clear ball.
    `extras`: additionally, sometimes, a thin-lens camera, file textures for albedo or roughness, a cut-out texture.
    For parity tests: scenes nobody arranged by hand."""
    rng = np.random.default_rng(seed)
    u = lambda lo, hi: float(rng.uniform(lo, hi))
    maps = list(_BASE_MAPS)
    media = list(_BASE_MEDIA)
    materials = list(_BASE_MATERIALS)
    foggy = rng.random() < 0.25
    if foggy:
        media.append(_medium("haze_medium", "HomogenousScatteringMedium", [u(0.02, 0.3)] * 3, [u(0, 0.05), u(0, 0.05), u(0, 0.05)], 1.0))
    media.append(_medium("murk_medium", "HomogenousScatteringMedium", [u(0.5, 4), u(0.5, 4), u(0.5, 4)], [u(0, 0.5), u(0, 0.5), u(0, 0.5)], 1.0,
                         emission=(u(0, 0.3), 0, 0)))
    outside = "haze_medium" if foggy else "air_medium"
    subtypes = ["Diffuse", "Reflective", "Transparent", "Glossy", "MetallicMicrofacet", "ReflectiveMicrofacet",
                "TransparentMicrofacet"]
    objects = []
    if rng.random() < 0.8:
        objects += _cornell_shell()
    else: # open scene: a ground plane and the background colour
        objects.append({"name": "ground", "type": "Object", "subtype": "Plane",
                        "parameters": {"position": [0, 0, 0], "v1": [1, 0, 0], "v2": [0, 0, -1], "material": "white_diffuse_material"}})
    n_own = int(rng.integers(4, 15))
    for i in range(n_own):
        name = f"m{i}"
        subtype = subtypes[int(rng.integers(0, len(subtypes)))]
        maps.append(_map3(name + "_albedo", [u(0.1, 1), u(0.1, 1), u(0.1, 1)]))
        maps.append(_map1(name + "_rough", u(0.02, 0.6)))
        par = {"albedo_map": name + "_albedo", "roughness_map": name + "_rough", "surface_refractive_index": u(1.1, 2.2)}
        if subtype in ("Reflective", "Glossy") and rng.random() < 0.5:
            par["is_dielectric"] = False
        if subtype == "MetallicMicrofacet":
            par["surface_refractive_index"] = u(0.2, 1.5)
            par["extinction_coefficient"] = u(1.5, 4)
        if subtype in ("Transparent", "TransparentMicrofacet"):
            inside = ["glass_medium", "cyan_glass_medium", "murk_medium", outside][int(rng.integers(0, 4))]
            par.update(internal_medium=inside, external_medium=outside)
        if rng.random() < 0.15:
            maps.append(_map1(name + "_veil", u(0.2, 0.8)))
            par["transparency_map"] = name + "_veil"
        materials.append(_material(name + "_material", subtype, **par))
        centre = [u(-0.9, 0.9), u(0.1, 1.6), u(-1.7, 0.6)]
        kind = rng.random()
        if kind < 0.55:
            objects.append(_sphere(name, centre, u(0.08, 0.4), name + "_material"))
        elif kind < 0.8:
            v1 = rng.normal(size=3)
            v2 = np.cross(v1, rng.normal(size=3))
            objects.append(_rect(name, centre, [float(x) for x in v1], [float(x) for x in v2], u(0.2, 0.9), u(0.2, 0.9), name + "_material"))
        else:
            pts = [[c + u(-0.4, 0.4) for c in centre] for _ in range(3)]
            objects.append({"name": name, "type": "Object", "subtype": "Triangle",
                            "parameters": {"p1": pts[0], "p2": pts[1], "p3": pts[2], "material": name + "_material"}})
    lights = int(rng.integers(1, 4)) # bit 0: rectangle light, bit 1: sphere light
    if lights & 1:
        objects.append(_rect("light_source", [u(-0.5, 0.5), 1.9999, u(-1.5, 0)], [1, 0, 0], [0, 0, 1], u(0.2, 0.8), u(0.2, 0.8), "light_source_material"))
    if lights & 2:
        maps += [_map1("bulb_strength", u(10, 60)), _map3("bulb_colour", [u(0.5, 1), u(0.5, 1), u(0.5, 1)])]
        materials.append(_material("bulb_material", "Diffuse", albedo_map="white", emission_color_map="bulb_colour",
                                   light_intensity_map="bulb_strength", is_light_source=True))
        objects.append(_sphere("bulb", [u(-0.7, 0.7), u(1.2, 1.7), u(-1.4, 0.2)], u(0.05, 0.15), "bulb_material"))
    scene = {"valuemaps": maps, "media": media, "materials": materials, "objects": objects, "background_medium": outside,
             "background_color": [u(0, 0.3), u(0, 0.3), u(0, 0.3)] if rng.random() < 0.5 else [0.0, 0.0, 0.0],
             "camera": _camera([u(-0.2, 0.2), u(0.7, 1.1), 1.9], u(82, 96), u(-6, 6), 0, 0.036, 0.036)}
    # zero to two OBJ meshes with one of the materials above, flat or smooth
    for k in range(int(rng.integers(0, 3))):
        os.makedirs(os.path.join(directory, "models"), exist_ok=True)
        write_bumpy_sphere_obj(os.path.join(directory, "models", f"blob{k}.obj"), int(rng.integers(6, 20)), int(rng.integers(4, 12)),
                               seed=int(rng.integers(1, 1000)), quads=bool(rng.random() < 0.5))
        material = materials[int(rng.integers(0, len(materials)))]["name"]
        objects.append({"name": f"blob{k}", "type": "Object", "subtype": "ObjectUnion",
                        "parameters": {"file": f"models/blob{k}.obj", "material": material,
                                       "enable_smooth_shading": bool(rng.random() < 0.5),
                                       "center": [u(-0.7, 0.7), u(0.3, 1.2), u(-1.5, 0.3)], "size": u(0.25, 0.7),
                                       "orientation": {"X": u(0, 90), "Y": u(0, 90), "Z": u(0, 90)}}})
    if extras: # drawn from a generator of their own, so that a seed gives the same base scene with or without them
        xr = np.random.default_rng(1_000_003 + seed)
        if xr.random() < 0.35:
            scene["camera"].update({"depth_of_field_mode": "circle" if xr.random() < 0.5 else "square",
                                    "focus_distance": float(xr.uniform(1.5, 3.5)), "aperture_size": float(xr.uniform(0.02, 0.15))})
        own = materials[len(_BASE_MATERIALS):len(_BASE_MATERIALS) + n_own] # the materials made above (the base ones are shared)
        for k in range(int(xr.integers(0, 3))): # file textures in place of constant maps
            if not own:
                break
            m = own[int(xr.integers(0, len(own)))]
            w, h = int(xr.integers(2, 40)), int(xr.integers(2, 40))
            if xr.random() < 0.6:
                write_map_file(os.path.join(directory, "maps", f"tex{k}.map"), value_noise(w, h, 3, seed=int(xr.integers(1, 999))))
                maps.append(_map_file(f"tex{k}", "ValueMap3D", f"maps/tex{k}.map"))
                m["parameters"]["albedo_map"] = f"tex{k}"
            else:
                write_map_file(os.path.join(directory, "maps", f"tex{k}.map"), 0.03 + 0.5 * value_noise(w, h, 1, seed=int(xr.integers(1, 999)))[:, :, 0])
                maps.append(_map_file(f"tex{k}", "ValueMap1D", f"maps/tex{k}.map"))
                m["parameters"]["roughness_map"] = f"tex{k}"
        if xr.random() < 0.25 and own: # a cut-out: transparency from a checker texture
            m = own[int(xr.integers(0, len(own)))]
            write_map_file(os.path.join(directory, "maps", "cut.map"), checker(16, 16, int(xr.integers(2, 8)), 0.0, float(xr.uniform(0.3, 1.0))))
            maps.append(_map_file("cut", "ValueMap1D", "maps/cut.map"))
            m["parameters"]["transparency_map"] = "cut"
    return write_scene_directory(directory, scene, default_settings(**settings))


def cloud_box(directory, fog=False, **settings):
    """The example box plus a homogeneous scattering volume bounded by an index-matched transparent
    sphere (which next-event estimation passes through), an emissive absorbing volume, and the
    coloured Beer's-law glass ball.  `fog=True` additionally makes the background medium scatter."""
    media = list(_BASE_MEDIA) + [
        _medium("cloud_medium", "HomogenousScatteringMedium", [2, 2, 2], [0.05, 0.05, 0.05], 1.0),
        _medium("ember_medium", "HomogenousScatteringMedium", [0.5, 0.2, 0.1], [1.0, 2.0, 3.0], 1.0,
                emission=[3.0, 1.0, 0.2]),
        _medium("fog_medium", "HomogenousScatteringMedium", [0.08, 0.08, 0.1], [0.01, 0.01, 0.0], 1.0),
    ]
    materials = list(_BASE_MATERIALS) + [
        _material("cloud_boundary_material", "Transparent", internal_medium="cloud_medium",
                  external_medium="fog_medium" if fog else "air_medium"),
        _material("ember_boundary_material", "Transparent", internal_medium="ember_medium",
                  external_medium="fog_medium" if fog else "air_medium"),
    ]
    objects = _cornell_shell() + [
        _sphere("blue_ball", [-0.35, 0.5, -1.3], 0.5, "blue_diffuse_material"),
        _sphere("glass_ball", [-0.5, 0.1, -0.5], 0.1, "cyan_glass_material"),
        _sphere("cloud", [0.35, 0.45, -0.6], 0.4, "cloud_boundary_material"),
        _sphere("ember", [-0.6, 0.2, 0.1], 0.2, "ember_boundary_material"),
        _sphere("gold_ball", [0.5, 0.25, -1.4], 0.25, "gold_material"),
        _LIGHT,
    ]
    scene = {"valuemaps": list(_BASE_MAPS), "media": media, "materials": materials, "objects": objects,
             "background_medium": "fog_medium" if fog else "air_medium", "background_color": [0.0, 0.0, 0.0],
             "camera": dict(_EXAMPLE_CAMERA)}
    return write_scene_directory(directory, scene, default_settings(**settings))


def emissive_mesh_box(directory, **settings):
    """Cornell walls lit only by a small emissive OBJ mesh (area-sampled triangle light)."""
    os.makedirs(directory, exist_ok=True)
    write_bumpy_sphere_obj(os.path.join(directory, "models", "lamp.obj"), 8, 5, seed=2, bump=0.2)
    materials = list(_BASE_MATERIALS)
    lamp = {"name": "lamp", "type": "Object", "subtype": "ObjectUnion",
            "parameters": {"file": "models/lamp.obj", "material": "light_source_material",
                           "center": [0.0, 1.5, -1.0], "size": 0.25}}
    objects = _cornell_shell() + [_sphere("blue_ball", [-0.35, 0.5, -1.3], 0.5, "blue_diffuse_material"),
                                  _sphere("gold_ball", [0.5, 0.25, -1], 0.25, "gold_material"), lamp]
    scene = {"valuemaps": list(_BASE_MAPS), "media": list(_BASE_MEDIA), "materials": materials, "objects": objects,
             "background_medium": "air_medium", "background_color": [0.0, 0.0, 0.0], "camera": dict(_EXAMPLE_CAMERA)}
    return write_scene_directory(directory, scene, default_settings(**settings))


def living_room(directory, mesh_lon=128, mesh_lat=65, texture_size=512, seed=11, **settings):
    """Stand-in for the reference's scenes/living_room, whose meshes and textures are not in its repository:
    the same kind of content -- 23 top-level objects, several OBJ meshes placed with center/size/orientation,
    file textures (value-noise albedo, checker roughness, an emission map), one small sphere light (radius 0.1)
    and a non-black background -- generated from `seed`."""
    os.makedirs(directory, exist_ok=True)
    n = texture_size
    write_map_file(os.path.join(directory, "maps", "floor_albedo.map"), 0.25 + 0.6 * value_noise(n, n, 3, seed=seed, cells=16))
    write_map_file(os.path.join(directory, "maps", "floor_roughness.map"), checker(n // 2, n // 2, 24, 0.05, 0.35))
    write_map_file(os.path.join(directory, "maps", "wall_albedo.map"), 0.5 + 0.4 * value_noise(n, n // 2, 3, seed=seed + 1, cells=6))
    write_map_file(os.path.join(directory, "maps", "fabric_albedo.map"), 0.15 + 0.7 * value_noise(n // 2, n // 2, 3, seed=seed + 2, cells=32))
    write_map_file(os.path.join(directory, "maps", "metal_roughness.map"), 0.08 + 0.3 * value_noise(n // 4, n // 4, 1, seed=seed + 3)[:, :, 0])
    screen = np.zeros((n // 4, n // 2, 3))
    screen[n // 16: n // 4 - n // 16, n // 16: n // 2 - n // 16] = value_noise(n // 2 - n // 8, n // 8, 3, seed=seed + 4, cells=4)
    write_map_file(os.path.join(directory, "maps", "screen_emission.map"), screen)
    maps = list(_BASE_MAPS) + [
        _map_file("floor_albedo", "ValueMap3D", "maps/floor_albedo.map"),
        _map_file("floor_roughness", "ValueMap1D", "maps/floor_roughness.map"),
        _map_file("wall_albedo", "ValueMap3D", "maps/wall_albedo.map"),
        _map_file("fabric_albedo", "ValueMap3D", "maps/fabric_albedo.map"),
        _map_file("metal_roughness", "ValueMap1D", "maps/metal_roughness.map"),
        _map_file("screen_emission", "ValueMap3D", "maps/screen_emission.map"),
        _map3("warm", [0.9922, 0.9569, 0.8627]), _map1("lamp_strength", 400), _map1("screen_strength", 3),
        _map1("wood_roughness", 0.3), _map3("grey", [0.55, 0.55, 0.55]),
    ]
    materials = list(_BASE_MATERIALS) + [
        _material("floor_material", "Glossy", albedo_map="floor_albedo", surface_refractive_index=1.3,
                  roughness_map="floor_roughness"),
        _material("wallpaper_material", "Diffuse", albedo_map="wall_albedo"),
        _material("fabric_material", "Diffuse", albedo_map="fabric_albedo"),
        _material("brushed_metal_material", "MetallicMicrofacet", albedo_map="grey", surface_refractive_index=1.5,
                  extinction_coefficient=3.2, roughness_map="metal_roughness"),
        _material("wood_material", "Glossy", albedo_map="fabric_albedo", surface_refractive_index=1.4,
                  roughness_map="wood_roughness"),
        _material("clear_glass_material", "Transparent", internal_medium="glass_medium", external_medium="air_medium"),
        _material("lamp_material", "Diffuse", albedo_map="white", emission_color_map="warm",
                  light_intensity_map="lamp_strength", is_light_source=True),
        _material("screen_material", "Diffuse", albedo_map="grey", emission_color_map="screen_emission",
                  light_intensity_map="screen_strength", is_light_source=True),
    ]
    models = os.path.join(directory, "models")
    write_bumpy_sphere_obj(os.path.join(models, "cushion.obj"), mesh_lon, mesh_lat, seed=seed, bump=0.05)
    write_bumpy_sphere_obj(os.path.join(models, "vase.obj"), mesh_lon, mesh_lat, seed=seed + 1, bump=0.12, quads=False)
    write_bumpy_sphere_obj(os.path.join(models, "ornament.obj"), mesh_lon // 2, mesh_lat // 2 + 1, seed=seed + 2, bump=0.2)

    def mesh(name, file, material, center, size, orientation, smooth=True):
        return {"name": name, "type": "Object", "subtype": "ObjectUnion",
                "parameters": {"file": "models/" + file, "material": material, "enable_smooth_shading": smooth,
                               "center": center, "size": size, "orientation": dict(zip("XYZ", orientation))}}

    objects = [
        _rect("floor", [0, 0, 0], [1, 0, 0], [0, 0, -1], 2.2, 4, "floor_material"),
        _rect("front_wall", [0, 1, -2], [1, 0, 0], [0, 1, 0], 2.2, 2, "wallpaper_material"),
        _rect("left_wall", [-1.1, 1, 0], [0, 0, -1], [0, 1, 0], 4, 2, "white_diffuse_material"),
        _rect("right_wall", [1.1, 1, 0], [0, 0, 1], [0, 1, 0], 4, 2, "wallpaper_material"),
        _rect("roof", [0, 2, 0], [1, 0, 0], [0, 0, 1], 2.2, 4, "white_diffuse_material"),
        _rect("rug", [0, 0.001, -0.8], [1, 0, 0], [0, 0, -1], 1.4, 1.2, "fabric_material"),
        _rect("table_top", [0.0, 0.35, -1.0], [1, 0, 0], [0, 0, -1], 0.8, 0.5, "wood_material"),
        _rect("shelf", [-1.0, 0.9, -1.2], [0, 0, -1], [1, 0, 0], 0.9, 0.2, "wood_material"),
        _rect("screen", [0.3, 1.1, -1.995], [1, 0, 0], [0, 1, 0], 0.9, 0.5, "screen_material"),
        _rect("window_pane", [1.09, 1.1, -0.6], [0, 0, 1], [0, 1, 0], 0.8, 0.8, "clear_glass_material"),
        mesh("cushion_a", "cushion.obj", "fabric_material", [-0.55, 0.18, -0.7], 0.3, (0, 20, 10)),
        mesh("cushion_b", "cushion.obj", "fabric_material", [0.6, 0.15, -0.3], 0.25, (15, 70, 0)),
        mesh("vase", "vase.obj", "clear_glass_material", [0.2, 0.52, -1.0], 0.16, (0, 0, 0)),
        mesh("metal_vase", "vase.obj", "brushed_metal_material", [-0.25, 0.5, -1.1], 0.14, (0, 45, 5), smooth=False),
        mesh("ornament", "ornament.obj", "gold_material", [-0.95, 1.02, -1.2], 0.1, (30, 0, 0)),
        _sphere("glass_ball", [0.05, 0.42, -0.85], 0.06, "cyan_glass_material"),
        _sphere("mirror_ball", [-0.7, 0.12, -0.1], 0.12, "mirror_material"),
        _sphere("rough_glass_ball", [0.75, 0.1, -1.2], 0.1, "rough_glass_material"),
        _sphere("blue_ball", [-0.9, 0.2, -1.6], 0.2, "blue_diffuse_material"),
        _sphere("gold_ball", [0.85, 0.45, -1.7], 0.12, "gold_material"),
        {"name": "table_leg", "type": "Object", "subtype": "Triangle",
         "parameters": {"p1": [-0.1, 0.0, -1.0], "p2": [0.1, 0.0, -1.0], "p3": [0.0, 0.35, -1.0], "material": "wood_material"}},
        _sphere("lamp", [-0.5, 1.7, -0.9], 0.1, "lamp_material"),
        _sphere("lamp_shade", [-0.5, 1.7, -0.9], 0.2, "clear_glass_material"),
    ]
    assert len(objects) == 23
    scene = {"valuemaps": maps, "media": list(_BASE_MEDIA), "materials": materials, "objects": objects,
             "background_medium": "air_medium", "background_color": [0.02, 0.03, 0.05],
             "camera": _camera([0, 0.9, 1.9], 88, 2, 0, 0.036, 0.036)}
    return write_scene_directory(directory, scene, default_settings(**settings))
