"""Host front end: settings / scene / denoising JSON, OBJ subset, .map textures, mmap'd buffers.
Behaviour follows the reference loaders (src/scene.cpp, src/objectunion.cpp:144-443,
src/valuemap.cpp:54-91, src/utils.cpp:245-280)."""
import json
import os

import numpy as np
import pytest

import pathtracing_amd as pt
from pathtracing_amd import _ffi, scenes


def test_default_settings_are_the_reference_defaults():
    s = pt.default_settings()
    assert (s.width, s.height, s.samples_per_pixel, s.samples_per_iteration) == (1000, 1000, 10, 16)
    assert (s.max_recursion_depth, s.min_recursion_steps, s.number_of_threads) == (100, 3, 1)
    assert (s.use_gamma_correction, s.bvh_leaf_size, s.bvh_n_axis_splits) == (1, 4, 16)
    assert (s.enable_next_event_estimation, s.enable_anti_aliasing) == (1, 1)


def test_settings_keys_and_validation(tmp_path):
    p = tmp_path / "settings.json"
    p.write_text(json.dumps({"WIDTH": 320, "HEIGHT": 200, "samples_per_pixel": 7, "enable_anti_aliasing": False,
                             "use_gamma_correction": False, "number_of_threads": 1}))
    s = pt.load_settings(str(p))
    assert (s.width, s.height, s.samples_per_pixel, s.enable_anti_aliasing, s.use_gamma_correction) == (320, 200, 7, 0, 0)
    assert s.samples_per_iteration == 16  # absent keys keep their defaults
    for bad, message in [({"samples_per_iteration": 0}, "samples_per_iteration"), ({"number_of_threads": 0}, "number_of_threads"),
                         ({"bvh_leaf_size": 0}, "bvh_leaf_size"), ({"bvh_n_axis_splits": 0}, "bvh_n_axis_splits")]:
        p.write_text(json.dumps(bad))
        with pytest.raises(pt.PtkError, match=message):
            pt.load_settings(str(p))
    with pytest.raises(pt.PtkError, match="Could not open file"):
        pt.load_settings(str(tmp_path / "missing.json"))


def test_example_scene_tables(tmp_path):
    d = scenes.cornell_example(str(tmp_path / "ex"), WIDTH=32, HEIGHT=16)
    hs = pt.load_scene_directory(d)
    desc = hs.desc
    assert desc.n_objects == 12 and desc.n_materials == 9 and desc.n_media == 3 and desc.n_maps == 9
    assert hs.object_names[0] == "floor" and hs.object_names[-1] == "light_source"
    kinds = [desc.objects[i].type for i in range(12)]
    assert kinds.count(_ffi.PTK_OBJ_RECTANGLE) == 7 and kinds.count(_ffi.PTK_OBJ_SPHERE) == 5
    light = desc.materials[desc.objects[11].material]
    assert light.is_light_source == 1 and light.type == _ffi.PTK_MAT_DIFFUSE
    # colour maps are converted sRGB -> linear at load when use_gamma_correction is on; 1-channel maps never
    red = desc.maps[0]
    assert abs(desc.map_data[red.offset] - ((0.8 + 0.055) / 1.055) ** 2.4) < 1e-15
    strength = [desc.maps[i] for i in range(9) if desc.maps[i].dim == 1][-1]
    assert desc.map_data[strength.offset] == 40.0
    assert desc.camera.orientation_x == 90 and desc.camera.dof_mode == 0
    hs2 = pt.load_scene_directory(d, use_gamma_correction=0)
    assert hs2.desc.map_data[hs2.desc.maps[0].offset] == 0.8


@pytest.mark.skipif(not os.path.isdir("/root/reference/scenes/example"), reason="reference tree not present")
def test_generated_example_equals_the_reference_scene_file(tmp_path):
    ours = pt.load_scene_directory(scenes.cornell_example(str(tmp_path / "ex"), WIDTH=1000, HEIGHT=1000))
    theirs = pt.load_scene_directory("/root/reference/scenes/example")
    a, b = ours.desc, theirs.desc
    assert ours.object_names == theirs.object_names
    assert bytes(a.camera) == bytes(b.camera)
    for i in range(a.n_objects):
        assert bytes(a.objects[i]) == bytes(b.objects[i])
    for i in range(a.n_materials):
        assert bytes(a.materials[i]) == bytes(b.materials[i])
    for i in range(a.n_media):
        assert bytes(a.media[i]) == bytes(b.media[i])
    assert [a.map_data[i] for i in range(a.n_map_values)] == [b.map_data[i] for i in range(b.n_map_values)]
    for key in ("samples_per_pixel", "samples_per_iteration", "max_recursion_depth", "min_recursion_steps"):
        assert getattr(ours.settings, key) == getattr(theirs.settings, key)


def _scene_with(objects, materials=None, maps=None):
    return {"valuemaps": maps or [], "media": [scenes._medium("air", "BeersLawMedium", [0, 0, 0], [0, 0, 0], 1.0)],
            "materials": materials or [scenes._material("m", "Diffuse")], "objects": objects, "background_medium": "air",
            "camera": scenes._camera([0, 0, 3], 90, 0, 0, 0.036, 0.036)}


def test_scene_errors_use_the_reference_messages():
    s = pt.default_settings()
    base = _scene_with([])
    for mutate, message in [
        (lambda d: d.pop("media"), "JSON structure missing key 'media'"),
        (lambda d: d.pop("background_medium"), "missing key 'background_medium'"),
        (lambda d: d.update(background_medium="fog"), "Medium entry 'fog' does not exist"),
        (lambda d: d["materials"].append(scenes._material("m", "Diffuse")), "Duplicate material entry 'm' found"),
        (lambda d: d["materials"].append(scenes._material("x", "Velvet")), "Velvet is not a valid material type"),
        (lambda d: d["objects"].append(scenes._sphere("s", [0, 0, 0], 1, "nope")), "Material entry 'nope' does not exist"),
        (lambda d: d["objects"].append({"name": "o", "type": "Object", "subtype": "Cube", "parameters": {"material": "m"}}),
         "Cube is not a valid object type"),
        (lambda d: d["objects"].append({"name": "o", "type": "Object", "subtype": "Sphere",
                                        "parameters": {"material": "m", "position": [0, 0], "radius": 1}}),
         "should contain 3 elements, but contains 2"),
        (lambda d: d["camera"].update(depth_of_field_mode="hexagon", focus_distance=1, aperture_size=0.1),
         "Invalid 'depth_of_field_mode' hexagon"),
    ]:
        d = json.loads(json.dumps(base))
        mutate(d)
        with pytest.raises(pt.PtkError, match=message):
            pt.HostScene.from_text(json.dumps(d), s)
    with pytest.raises(pt.PtkError, match="JSON parse error"):
        pt.HostScene.from_text("{ not json", s)


def _load_obj(tmp_path, text, **params):
    (tmp_path / "m.obj").write_text(text)
    obj = {"name": "mesh", "type": "Object", "subtype": "ObjectUnion", "parameters": dict(file="m.obj", material="m", **params)}
    hs = pt.HostScene.from_text(json.dumps(_scene_with([obj])), pt.default_settings(), str(tmp_path))
    n = hs.desc.n_triangles
    v = np.array([hs.desc.tri_vertices[i] for i in range(9 * n)]).reshape(n, 3, 3)
    flags = [hs.desc.tri_flags[i] for i in range(n)]
    return hs, v, flags


def test_obj_subset_quads_spacing_and_index_forms(tmp_path):
    text = ("# comment\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nvn 0 0 1\n"
            "f 1 2 3 4\n"            # quad -> (1,2,3) + (1,3,4)
            "f 1/1 2/2 3/3\n"        # v/vt
            "f 1//1 2//1 3//1\n"     # v//vn
            "f 1/1/1 2/2/1 3/3/1\n"  # v/vt/vn
            "f 1 2  3\n"             # double space: 4 spaces but an empty field -> dropped like the reference
            "f 1 2 3 4 1\n")         # five corners: silently dropped
    hs, v, flags = _load_obj(tmp_path, text, enable_smooth_shading=True)
    assert len(v) == 5
    assert np.array_equal(v[0], [[0, 0, 0], [1, 0, 0], [1, 1, 0]]) and np.array_equal(v[1], [[0, 0, 0], [1, 1, 0], [0, 1, 0]])
    assert flags == [0, 0, _ffi.PTK_TRI_HAS_UV, _ffi.PTK_TRI_SMOOTH, _ffi.PTK_TRI_HAS_UV | _ffi.PTK_TRI_SMOOTH]
    uv = [hs.desc.tri_uv[6 * 2 + i] for i in range(6)]
    assert uv == [0, 0, 1, 0, 1, 1]
    hs2, _, flags2 = _load_obj(tmp_path, text)  # smooth shading off: normals ignored
    assert flags2 == [0, 0, _ffi.PTK_TRI_HAS_UV, 0, _ffi.PTK_TRI_HAS_UV]


def test_obj_transform_applies_only_with_a_center(tmp_path):
    text = "v -2 0 0\nv 2 0 0\nv 0 4 0\nf 1 2 3\n"
    _, v, _ = _load_obj(tmp_path, text, size=3.0)  # no center: untouched
    assert np.array_equal(v[0], [[-2, 0, 0], [2, 0, 0], [0, 4, 0]])
    _, v, _ = _load_obj(tmp_path, text, center=[10, 0, 0], size=3.0)
    mid = np.array([0, 2, 0.0])
    far = max(np.linalg.norm(p - mid) for p in np.array([[-2, 0, 0], [2, 0, 0], [0, 4, 0.0]]))
    want = (np.array([[-2, 0, 0], [2, 0, 0], [0, 4, 0.0]]) - mid) / far * 3.0 + np.array([10, 0, 0.0])
    np.testing.assert_allclose(v[0], want, rtol=1e-15, atol=1e-15)
    _, v, _ = _load_obj(tmp_path, text, center=[0, 0, 0], orientation={"X": 0, "Y": 90, "Z": 0})
    np.testing.assert_allclose(v[0][1], [0, -2, -2], atol=1e-12)  # (2,-2,0) rotated 90 degrees about y


def test_obj_parsed_in_parts_gives_the_same_triangles_and_the_first_error(tmp_path, monkeypatch):
    """A generated mesh parsed whole and cut into 2, 7 and 64 parts (more parts than some sections have lines): identical
    arrays.  With two bad lines in the file, the error is the one of the first bad line whichever part meets which."""
    d = str(tmp_path / "mesh")
    n = scenes.mesh_box(d, n_lon=48, n_lat=25, material="gold", WIDTH=16, HEIGHT=16)
    def arrays():
        hs = pt.load_scene_directory(d)
        desc = hs.desc
        assert desc.n_triangles == n
        return [np.ctypeslib.as_array(a, shape=(n * k,)).copy() for a, k in ((desc.tri_vertices, 9), (desc.tri_uv, 6), (desc.tri_normals, 9))] + \
               [np.ctypeslib.as_array(desc.tri_flags, shape=(n,)).copy()]
    whole = arrays()
    for parts in (2, 7, 64):
        monkeypatch.setenv("PTK_OBJ_PARSE_PARTS", str(parts))
        for a, b in zip(whole, arrays()):
            assert np.array_equal(a, b), parts
    path = os.path.join(d, "models", "bumpy_sphere.obj")
    lines = open(path).read().split("\n")
    vertex_lines = [i for i, l in enumerate(lines) if l.startswith("v ")]
    lines[vertex_lines[10]] = "v first_bad 0 0"
    lines[vertex_lines[-10]] = "v second_bad 0 0"
    open(path, "w").write("\n".join(lines))
    for parts in (1, 2, 7, 64):
        monkeypatch.setenv("PTK_OBJ_PARSE_PARTS", str(parts))
        with pytest.raises(pt.PtkError) as e:
            pt.load_scene_directory(d)
        assert "first_bad" in str(e.value), parts


def test_obj_errors(tmp_path):
    with pytest.raises(pt.PtkError, match="Could not open file 'm2.obj'"):
        obj = {"name": "mesh", "type": "Object", "subtype": "ObjectUnion", "parameters": {"file": "m2.obj", "material": "m"}}
        pt.HostScene.from_text(json.dumps(_scene_with([obj])), pt.default_settings(), str(tmp_path))
    with pytest.raises(pt.PtkError, match="out of range"):
        _load_obj(tmp_path, "v 0 0 0\nv 1 0 0\nf 1 2 3\n")


def test_map_files_round_trip_and_validation(tmp_path):
    tex = np.arange(2 * 3 * 3, dtype=np.float64).reshape(2, 3, 3) / 18.0
    pt.write_map(str(tmp_path / "a.map"), tex)
    raw = np.fromfile(str(tmp_path / "a.map"))
    assert list(raw[:3]) == [3, 2, 3] and np.array_equal(raw[3:], tex.ravel())
    maps = [{"name": "a", "type": "ValueMap3D", "parameters": {"file": "a.map"}}]
    scene = _scene_with([scenes._sphere("s", [0, 0, 0], 1, "m")], [scenes._material("m", "Diffuse", albedo_map="a")], maps)
    s = pt.default_settings()
    s.use_gamma_correction = 0
    hs = pt.HostScene.from_text(json.dumps(scene), s, str(tmp_path))
    m = hs.desc.maps[0]
    assert (m.width, m.height, m.dim) == (3, 2, 3)
    assert [hs.desc.map_data[m.offset + i] for i in range(18)] == list(tex.ravel())
    s.use_gamma_correction = 1
    hs = pt.HostScene.from_text(json.dumps(scene), s, str(tmp_path))
    x = tex.ravel()[5]
    assert abs(hs.desc.map_data[m.offset + 5] - ((x + 0.055) / 1.055) ** 2.4) < 1e-15
    np.array([3.0, 2.0, 3.0, 1.0]).tofile(str(tmp_path / "bad.map"))
    scene["valuemaps"][0]["parameters"]["file"] = "bad.map"
    with pytest.raises(pt.PtkError, match="does not follow the expected format"):
        pt.HostScene.from_text(json.dumps(scene), s, str(tmp_path))


def test_denoising_pipeline_and_validation(tmp_path):
    s = pt.default_settings()
    s.width, s.height = 64, 48
    p = tmp_path / "denoising.json"
    p.write_text(json.dumps({"pipeline": [
        {"mode": "median", "parameters": {"kernel_size": 3, "threshold": 0.5}},
        {"mode": "skip", "parameters": {}},
        {"mode": "atrous", "parameters": {"iterations": 4, "sigma_rt": 2, "sigma_x": 2, "sigma_n": 1}}]}))
    tasks = pt.load_denoising(str(p), s)
    assert len(tasks) == 2 and tasks[0].mode == _ffi.PTK_DENOISE_MEDIAN and tasks[0].kernel_size == 3
    assert tasks[1].iterations == 4 and tasks[1].sigma_n == 1.0
    for pipeline, message in [([{"mode": "atrous", "parameters": {"iterations": 6, "sigma_rt": 1, "sigma_x": 1, "sigma_n": 1}}],
                               "Too many denoising iterations"),
                              ([{"mode": "median", "parameters": {"kernel_size": 49, "threshold": 0}}], "kernel size is too large"),
                              ([{"mode": "median", "parameters": {"kernel_size": 3, "threshold": -1}}], "cannot be negative"),
                              ([{"mode": "blur", "parameters": {}}], "'blur' is not a valid denoising mode"),
                              ([{"mode": "median", "parameters": {"kernel_size": 3}}], "missing key 'threshold'")]:
        p.write_text(json.dumps({"pipeline": pipeline}))
        with pytest.raises(pt.PtkError, match=message):
            pt.load_denoising(str(p), s)


def test_accumulation_buffer_files(tmp_path):
    """temp/raw_*.dat: float64 [H][W][3], shared mapping, truncated on request (src/utils.cpp:245-273)."""
    import ctypes as C
    lib = pt.host_lib()
    path = str(tmp_path / "raw_pixel.dat").encode()
    size = 4 * 3 * 3 * 8
    fd = C.c_int()
    ptr = lib.ptk_host_map_buffer(path, size, 1, C.byref(fd))
    assert ptr
    a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(3, 4, 3))
    assert not a.any()
    a[1, 2] = [1.5, 2.5, 3.5]
    lib.ptk_host_unmap_buffer(ptr, size, fd)
    b = np.fromfile(path.decode()).reshape(3, 4, 3)
    assert list(b[1, 2]) == [1.5, 2.5, 3.5] and os.stat(path.decode()).st_mode & 0o777 == 0o600
    ptr = lib.ptk_host_map_buffer(path, size, 0, C.byref(fd))  # re-open without clearing, as the denoiser does
    a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(3, 4, 3))
    assert list(a[1, 2]) == [1.5, 2.5, 3.5]
    lib.ptk_host_unmap_buffer(ptr, size, fd)


def test_mutated_scene_directories_load_or_fail_with_a_message():
    """tools/fuzz_loader.py in a process of its own (a crash of the loader would take the test runner with it)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_loader.py"), "0", "120"], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    assert "refused" in run.stdout.splitlines()[-1]


# ---- the loaders against the reference's own --------------------------------------------------------------------
# tests/golden/loader_obj.npz and loader_map.npz hold what the reference's load_object_model and create_value_map
# (src/objectunion.cpp:425-443, src/valuemap.cpp:54-100) made of the inputs in tests/golden/loader_cases.py
# (generated by tests/golden/make_golden.py from oracle/_ref/libref.so).  This repository's loader must produce the same
# numbers, bit for bit, and fail where the reference fails.
import sys  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import loader_cases  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def loader_inputs(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("loader_inputs"))
    return (d,) + loader_cases.write_inputs(d)


@pytest.mark.parametrize("parts", [0, 5], ids=["whole", "in-5-parts"])
@pytest.mark.parametrize("name,key,params", loader_cases.OBJ_CASES, ids=[c[0] for c in loader_cases.OBJ_CASES])
def test_obj_loader_matches_the_reference_loader(loader_inputs, monkeypatch, name, key, params, parts):
    """parts = 5: the file is cut at line ends into five parts that are parsed on threads of their own and joined in
    file order (what files of a megabyte or more get, one part per hardware thread); the result must not depend on it."""
    if parts:
        monkeypatch.setenv("PTK_OBJ_PARSE_PARTS", str(parts))
    directory, obj_paths, _ = loader_inputs
    golden = np.load(os.path.join(GOLDEN, "loader_obj.npz"))
    file = os.path.basename(obj_paths[key]) if key else "does_not_exist.obj"
    obj = {"name": "mesh", "type": "Object", "subtype": "ObjectUnion", "parameters": dict(file=file, material="m", **params)}
    text = json.dumps(_scene_with([obj]))
    if name + "_error" in golden:  # the reference throws: so must the loader (the wording is the reference's for a missing file)
        with pytest.raises(pt.PtkError) as e:
            pt.HostScene.from_text(text, pt.default_settings(), directory)
        if "Could not open" in str(golden[name + "_error"]):
            assert "Could not open file" in str(e.value)
        return
    hs = pt.HostScene.from_text(text, pt.default_settings(), directory)
    d = hs.desc
    n = d.n_triangles
    want = golden[name + "_triangles"]
    assert n == len(want)
    if n == 0:
        return
    v = np.ctypeslib.as_array(d.tri_vertices, shape=(n, 9))
    uv = np.ctypeslib.as_array(d.tri_uv, shape=(n, 6))
    nrm = np.ctypeslib.as_array(d.tri_normals, shape=(n, 9)).copy()
    flags = np.ctypeslib.as_array(d.tri_flags, shape=(n,))
    smooth = (flags & _ffi.PTK_TRI_SMOOTH) != 0
    # vertex normals matter for smooth triangles only: a flat one carries its face normal in the reference and
    # unspecified values here (the device library derives the face normal itself)
    nrm[~smooth] = 0.0
    want = want.copy()
    want[want[:, 24] == 0, 15:24] = 0.0
    want = want[np.lexsort(want.T[::-1])]
    assert (uv[(flags & _ffi.PTK_TRI_HAS_UV) == 0] == 0).all()
    rows = np.concatenate([v, uv, nrm, smooth[:, None].astype(np.float64)], axis=1)
    rows = rows[np.lexsort(rows.T[::-1])]  # the reference's array is in the order its hierarchy's root sorted it into
    assert np.array_equal(rows, want), name


@pytest.mark.parametrize("name,source,channels,gamma", loader_cases.MAP_CASES, ids=[c[0] for c in loader_cases.MAP_CASES])
def test_map_loader_matches_the_reference_loader(loader_inputs, name, source, channels, gamma):
    directory, _, map_paths = loader_inputs
    golden = np.load(os.path.join(GOLDEN, "loader_map.npz"))
    kind = "ValueMap1D" if channels == 1 else "ValueMap3D"
    maps = [scenes._map_file("tex", kind, os.path.basename(map_paths[name]))]
    material = scenes._material("m", "Diffuse", **({"roughness_map": "tex"} if channels == 1 else {"albedo_map": "tex"}))
    text = json.dumps(_scene_with([scenes._sphere("s", [0, 0, 0], 1, "m")], materials=[material], maps=maps))
    settings = pt.default_settings()
    settings.use_gamma_correction = 1 if gamma else 0
    if name + "_error" in golden:
        with pytest.raises(pt.PtkError) as e:
            pt.HostScene.from_text(text, settings, directory)
        reference_message = str(golden[name + "_error"]).replace("<file>", os.path.basename(map_paths[name]))
        assert reference_message in str(e.value).replace(directory + os.sep, "")
        return
    hs = pt.HostScene.from_text(text, settings, directory)
    m = hs.desc.maps[0]
    want = golden[name + "_data"]
    assert (m.height, m.width, m.dim) == want.shape
    got = np.ctypeslib.as_array(hs.desc.map_data, shape=(hs.desc.n_map_values,))[m.offset:m.offset + want.size]
    assert np.array_equal(got.reshape(want.shape), want), name
