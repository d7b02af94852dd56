"""GPU parity: the HIP library (through its C ABI) against the CPU oracle on the same scenes and the
same counter-based random streams.  Tolerances: ids bit-exact except where a ray grazes an edge
(>= 99.99 % agreement required); fp64 radiance per sample within 1e-9 relative for >= 99.9 % of the
samples (device libm and fused multiply-adds differ from the host's in the last bits, and a path
whose discrete decision sits within that noise takes another branch), image means within 2e-3 relative."""
import numpy as np
import pytest

import pathtracing_amd as pt
from pathtracing_amd.distributed import band_job

pytestmark = pytest.mark.gpu

SCENES = ["example", "zoo", "zoo_circle", "zoo_square", "mesh_gold", "mesh_rough_glass", "mesh_glass",
          "mesh_textured", "cloud", "fog", "lamp", "thicket", "room"]


@pytest.fixture(scope="module")
def ob():
    from oracle import binding
    return binding


@pytest.fixture(scope="module", autouse=True)
def gpu():
    pt.init(0)


def _close_fraction(a, b, rtol=1e-9, atol=1e-12):
    ok = np.isclose(a, b, rtol=rtol, atol=atol, equal_nan=True)
    return ok.reshape(ok.shape[0], -1).all(axis=1).mean()


@pytest.mark.parametrize("name", SCENES)
def test_primary_hits_match_oracle(scene_dirs, ob, name):
    hs = pt.load_scene_directory(scene_dirs[name], enable_anti_aliasing=0)
    n = hs.settings.width * hs.settings.height
    job = pt.make_job(0, n, 1, 0, 3)
    want = ob.OracleScene(hs).trace_primary(job)
    got = pt.DeviceScene(hs).trace_primary(job)
    same = (want[0] == got[0]) & (want[1] == got[1])
    assert same.mean() >= 0.9999, f"id agreement {same.mean()}"
    hit = same & (want[0] >= 0)
    np.testing.assert_allclose(got[2][hit], want[2][hit], rtol=1e-11)
    np.testing.assert_allclose(got[3][hit], want[3][hit], rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(got[4][hit], want[4][hit], rtol=1e-9, atol=1e-11)


# tail_records: with the default the small test frames are traced by k_tail alone (one lane follows one
# path to its end); 0 sends every bounce through the class sort and k_bounce (and, with meshes, the
# deferral, k_traverse and k_bounce_resume); 1500 switches from the one to the other part-way.
@pytest.mark.parametrize("tail_records", [1048576, 0, 1500])
@pytest.mark.parametrize("name", SCENES)
def test_samples_match_oracle(scene_dirs, ob, name, tail_records):
    hs = pt.load_scene_directory(scene_dirs[name])
    n = hs.settings.width * hs.settings.height
    job = pt.make_job(0, n, 6, 2, 12345)
    want = ob.OracleScene(hs).trace_samples(job)
    pt.set_option("tail_records", tail_records)
    try:
        got = pt.DeviceScene(hs).trace_samples(job)
    finally:
        pt.set_option("tail_records", 1048576)
    for w, g, what in zip(want, got, ("radiance", "position", "normal")):
        w2, g2 = w.reshape(-1, 3), g.reshape(-1, 3)
        frac = _close_fraction(g2, w2)
        assert frac >= 0.999, f"{name} {what}: only {frac:.5f} of the samples agree"
    mean_w, mean_g = want[0].mean(axis=(0, 1)), got[0].mean(axis=(0, 1))
    np.testing.assert_allclose(mean_g, mean_w, rtol=2e-3)


@pytest.mark.parametrize("tail_records", [1048576, 0])
# 74, 88, 96: an emissive mesh next to an opaque one (a connection to the former must not stop at the first triangle
# of the latter it meets, which may lie beyond the light)
@pytest.mark.parametrize("seed", list(range(12)) + [74, 88, 96])
def test_random_scenes_match_oracle(tmp_path, ob, seed, tail_records):
    """Scenes nobody arranged by hand (scenes.random_box: random objects, materials of every type, media, lights; odd seeds
    also file textures, cut-outs and thin-lens cameras);
    the oracle is bit-exact against the reference on the same twelve scenes (tests/test_oracle_golden.py)."""
    from pathtracing_amd import scenes
    hs = pt.load_scene_directory(scenes.random_box(str(tmp_path), seed, extras=bool(seed % 2), WIDTH=48, HEIGHT=32))
    n = hs.settings.width * hs.settings.height
    osc, dev = ob.OracleScene(hs), pt.DeviceScene(hs)
    primary = pt.make_job(0, n, 1, 0, 3)
    want, got = osc.trace_primary(primary), dev.trace_primary(primary)
    assert ((want[0] == got[0]) & (want[1] == got[1])).mean() >= 0.999
    job = pt.make_job(0, n, 6, 1, 99 + seed)
    want = osc.trace_samples(job)
    pt.set_option("tail_records", tail_records)
    try:
        got = dev.trace_samples(job)
    finally:
        pt.set_option("tail_records", 1048576)
    for w, g, what in zip(want, got, ("radiance", "position", "normal")):
        frac = _close_fraction(g.reshape(-1, 3), w.reshape(-1, 3))
        assert frac >= 0.999, f"seed {seed} {what}: only {frac:.5f} of the samples agree"
    np.testing.assert_allclose(got[0].mean(axis=(0, 1)), want[0].mean(axis=(0, 1)), rtol=2e-3)


@pytest.mark.parametrize("name", ["example", "mesh_rough_glass", "cloud"])
def test_results_do_not_depend_on_batch_size_or_knobs(scene_dirs, name):
    hs = pt.load_scene_directory(scene_dirs[name])
    n = hs.settings.width * hs.settings.height
    job = pt.make_job(0, n, 4, 0, 7)
    whole = pt.DeviceScene(hs).trace_samples(job)
    pieces = pt.DeviceScene(hs, max_paths_in_flight=1000).trace_samples(job)
    for a, b in zip(whole, pieces):
        assert np.array_equal(a, b)
    for tail_records in (0, 700):
        pt.set_option("tail_records", tail_records)
        try:
            other = pt.DeviceScene(hs).trace_samples(job)
        finally:
            pt.set_option("tail_records", 1048576)
        for a, b in zip(whole, other):
            assert np.array_equal(a, b)


# specialised_kernels = 0: every scene runs the general kernel variants (all features compiled in, scene tables read
# from memory instead of their LDS copies), which is what scenes with more objects than the LDS copies hold get.
@pytest.mark.parametrize("name", SCENES)
def test_general_kernels_give_the_same_results(scene_dirs, name):
    hs = pt.load_scene_directory(scene_dirs[name])
    n = hs.settings.width * hs.settings.height
    job = pt.make_job(0, n, 3, 0, 31)
    for tail_records in (1048576, 0):
        pt.set_option("tail_records", tail_records)
        try:
            dev = pt.DeviceScene(hs)
            specialised = dev.trace_samples(job)
            assert dev.bounce_kernel().endswith("true>"), dev.bounce_kernel()
            pt.set_option("specialised_kernels", 0)
            assert dev.bounce_kernel().endswith("true, false>"), dev.bounce_kernel()
            general = dev.trace_samples(job)
        finally:
            pt.set_option("specialised_kernels", 1)
            pt.set_option("tail_records", 1048576)
        for a, b in zip(specialised, general):
            assert np.array_equal(a, b)


def test_pixel_subrange_equals_full_frame(scene_dirs):
    hs = pt.load_scene_directory(scene_dirs["zoo"])
    n = hs.settings.width * hs.settings.height
    dev = pt.DeviceScene(hs)
    full = dev.trace_samples(pt.make_job(0, n, 3, 5, 99))[0]
    part = dev.trace_samples(pt.make_job(1000, 500, 3, 5, 99))[0]
    assert np.array_equal(full[1000:1500], part)


def test_film_running_mean_matches_oracle(scene_dirs, ob):
    hs = pt.load_scene_directory(scene_dirs["example"])
    W, H = hs.settings.width, hs.settings.height
    n = W * H
    dev = pt.DeviceScene(hs)
    film = pt.Film(W, H)
    osc = ob.OracleScene(hs)
    img, pos, nrm = np.zeros((H, W, 3)), np.zeros((H, W, 3)), np.zeros((H, W, 3))
    cnt = np.zeros((H, W), dtype=np.uint64)
    for start, count, spp, offset in [(0, n, 4, 0), (0, n // 2, 3, 4), (n // 2, n - n // 2, 5, 4)]:
        job = pt.make_job(start, count, spp, offset, 21)
        dev.render(film, job)
        osc.render(job, img, pos, nrm, cnt, threads=4)
    g_img, g_pos, g_nrm, g_cnt = film.download()
    assert np.array_equal(g_cnt, cnt)
    assert _close_fraction(g_img.reshape(-1, 3), img.reshape(-1, 3), rtol=1e-9) >= 0.99
    np.testing.assert_allclose(g_img.mean(), img.mean(), rtol=1e-4)
    np.testing.assert_allclose(g_pos, pos, rtol=1e-9, atol=1e-9)
    assert _close_fraction(g_nrm.reshape(-1, 3), nrm.reshape(-1, 3), rtol=1e-8, atol=1e-10) >= 0.999


def test_render_host_updates_caller_buffers(scene_dirs):
    hs = pt.load_scene_directory(scene_dirs["example"])
    W, H = hs.settings.width, hs.settings.height
    n = W * H
    dev = pt.DeviceScene(hs)
    film = pt.Film(W, H)
    img, pos, nrm = np.zeros((H, W, 3)), np.zeros((H, W, 3)), np.zeros((H, W, 3))
    cnt = np.zeros((H, W), dtype=np.uint64)
    for start, count in [(0, 1500), (1500, n - 1500)]:
        job = pt.make_job(start, count, 2, 0, 5)
        dev.render_host(job, img, pos, nrm, cnt)
        dev.render(film, job)
    g = film.download()
    assert np.array_equal(g[0], img) and np.array_equal(g[1], pos) and np.array_equal(g[2], nrm)
    assert np.array_equal(g[3], cnt) and cnt.min() == 2


def test_denoise_matches_oracle(scene_dirs, ob):
    hs = pt.load_scene_directory(scene_dirs["example"])
    W, H = hs.settings.width, hs.settings.height
    dev = pt.DeviceScene(hs)
    film = pt.Film(W, H)
    dev.render(film, pt.make_job(0, W * H, 8, 0, 1))
    img, pos, nrm, _ = film.download()
    for pipeline in ([{"mode": "median", "kernel_size": 3, "threshold": 0.5},
                      {"mode": "atrous", "iterations": 4, "sigma_rt": 2, "sigma_x": 2, "sigma_n": 2}],
                     [{"mode": "atrous", "iterations": 2, "sigma_rt": 0.7, "sigma_x": 1.5, "sigma_n": 0.4}],
                     [{"mode": "median", "kernel_size": 5, "threshold": 0.05}],
                     [{"mode": "median", "kernel_size": 7, "threshold": 0.0}]):
        tasks = pt.make_denoise_tasks(pipeline)
        want = ob.oracle_denoise(tasks, img, pos, nrm)
        got = pt.denoise(tasks, img, pos, nrm)
        np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-13)


def test_random_denoise_pipelines_match_oracle(ob):
    """Random frames (1x1 to 89x69, fireflies, zero and NaN pixels) through random pipelines (tools/random_denoise.py):
    equal to the oracle within 1e-10, and a pipeline whose taps would leave a small frame is refused."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "random_denoise", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "random_denoise.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    verdicts = {seed: tool.check_case(seed, pt, ob) for seed in range(1, 81)}
    assert all(v in ("ok", "refused") for v in verdicts.values()), {s: v for s, v in verdicts.items() if v not in ("ok", "refused")}
    assert sum(v == "refused" for v in verdicts.values()) >= 5 and sum(v == "ok" for v in verdicts.values()) >= 40


def test_statistics_count_paths_and_rays(scene_dirs, ob):
    hs = pt.load_scene_directory(scene_dirs["example"])
    n = hs.settings.width * hs.settings.height
    job = pt.make_job(0, n, 4, 0, 3)
    dev = pt.DeviceScene(hs)
    dev.trace_samples(job)
    osc = ob.OracleScene(hs)
    osc.trace_samples(job)
    s, o = dev.stats(), osc.stats()
    assert s.paths == o.paths == n * 4
    assert abs(int(s.extension_rays) - int(o.extension_rays)) <= 0.001 * o.extension_rays
    # the GPU does not march connections that face away from the surface (they contribute nothing);
    # the reference marches them before discarding them
    assert 0.85 * o.shadow_rays <= s.shadow_rays <= o.shadow_rays
    assert s.algorithmic_bytes > 0


@pytest.mark.parametrize("overrides", [
    dict(enable_next_event_estimation=0),
    dict(enable_anti_aliasing=0),
    dict(max_recursion_depth=0),
    dict(max_recursion_depth=1, min_recursion_steps=0),
    dict(min_recursion_steps=0),
    dict(min_recursion_steps=50, max_recursion_depth=12),
], ids=lambda d: ",".join(f"{k}={v}" for k, v in d.items()))
@pytest.mark.parametrize("name", ["example", "zoo", "mesh_rough_glass", "fog"])
def test_settings_variations_match_oracle(scene_dirs, ob, name, overrides):
    hs = pt.load_scene_directory(scene_dirs[name], **overrides)
    n = hs.settings.width * hs.settings.height
    job = pt.make_job(0, n, 3, 1, 4242)
    want = ob.OracleScene(hs).trace_samples(job)
    for tail_records in (1048576, 0):
        pt.set_option("tail_records", tail_records)
        try:
            got = pt.DeviceScene(hs).trace_samples(job)
        finally:
            pt.set_option("tail_records", 1048576)
        for w, g, what in zip(want, got, ("radiance", "position", "normal")):
            frac = _close_fraction(g.reshape(-1, 3), w.reshape(-1, 3))
            assert frac >= 0.999, f"{name} {overrides} {what}: only {frac:.5f} of the samples agree"


def test_scene_with_more_objects_than_the_on_chip_tables_hold(tmp_path, ob):
    """150 objects do not fit the LDS copies of the scene tables: the scene is traced by the general kernels."""
    from pathtracing_amd import scenes
    rng = np.random.default_rng(3)
    scene = scenes.cornell_example_dict()
    materials = [m["name"] for m in scene["materials"]]
    for k in range(138):
        position = [float(rng.uniform(-0.8, 0.8)), float(rng.uniform(0.05, 1.5)), float(rng.uniform(-1.8, -0.2))]
        scene["objects"].append(scenes._sphere(f"bead_{k}", position, float(rng.uniform(0.02, 0.06)), materials[k % len(materials)]))
    d = scenes.write_scene_directory(str(tmp_path / "beads"), scene, scenes.default_settings(WIDTH=48, HEIGHT=36))
    hs = pt.load_scene_directory(d)
    n = hs.settings.width * hs.settings.height
    job = pt.make_job(0, n, 4, 0, 9)
    want = ob.OracleScene(hs).trace_samples(job)
    for tail_records in (1048576, 0):
        pt.set_option("tail_records", tail_records)
        try:
            dev = pt.DeviceScene(hs)
            assert dev.bounce_kernel().endswith("false>"), dev.bounce_kernel()
            got = dev.trace_samples(job)
        finally:
            pt.set_option("tail_records", 1048576)
        for w, g, what in zip(want, got, ("radiance", "position", "normal")):
            frac = _close_fraction(g.reshape(-1, 3), w.reshape(-1, 3))
            assert frac >= 0.999, f"{what}: only {frac:.5f} of the samples agree"


def test_scene_with_more_meshes_than_the_traversal_kernel_keeps_on_chip(tmp_path, ob):
    """270 small meshes: the list of meshes is read from memory by k_traverse, and the general kernels shade."""
    import os
    from pathtracing_amd import scenes
    rng = np.random.default_rng(5)
    d = str(tmp_path / "swarm")
    scenes.write_bumpy_sphere_obj(os.path.join(d, "models", "pebble.obj"), 8, 5, seed=3, bump=0.15)
    scene = scenes.cornell_example_dict()
    materials = ["red_diffuse_material", "gold_material", "white_diffuse_material", "blue_diffuse_material"]
    for k in range(270):
        center = [float(rng.uniform(-0.8, 0.8)), float(rng.uniform(0.1, 1.6)), float(rng.uniform(-1.8, -0.2))]
        scene["objects"].append({"name": f"pebble_{k}", "type": "Object", "subtype": "ObjectUnion",
                                 "parameters": {"file": "models/pebble.obj", "material": materials[k % 4],
                                                "enable_smooth_shading": bool(k & 1), "center": center,
                                                "size": float(rng.uniform(0.04, 0.09)),
                                                "orientation": {"X": float(rng.uniform(0, 90)), "Y": 0.0, "Z": 0.0}}})
    scenes.write_scene_directory(d, scene, scenes.default_settings(WIDTH=40, HEIGHT=30))
    hs = pt.load_scene_directory(d)
    n = hs.settings.width * hs.settings.height
    job = pt.make_job(0, n, 3, 0, 21)
    want = ob.OracleScene(hs).trace_samples(job)
    for tail_records in (1048576, 0):
        pt.set_option("tail_records", tail_records)
        try:
            dev = pt.DeviceScene(hs)
            assert dev.bounce_kernel().endswith("false>"), dev.bounce_kernel()
            got = dev.trace_samples(job)
        finally:
            pt.set_option("tail_records", 1048576)
        for w, g, what in zip(want, got, ("radiance", "position", "normal")):
            frac = _close_fraction(g.reshape(-1, 3), w.reshape(-1, 3))
            assert frac >= 0.999, f"{what}: only {frac:.5f} of the samples agree"


def test_degenerate_jobs_and_scenes(tmp_path, ob):
    from pathtracing_amd import scenes
    # a scene with no light and one with nothing but the background
    scene = scenes.cornell_example_dict()
    scene["objects"] = [o for o in scene["objects"] if o["name"] != "light_source"]
    dark = scenes.write_scene_directory(str(tmp_path / "dark"), scene, scenes.default_settings(WIDTH=32, HEIGHT=24))
    scene = scenes.cornell_example_dict()
    scene["objects"] = []
    scene["background_color"] = [0.25, 0.5, 0.75]
    empty = scenes.write_scene_directory(str(tmp_path / "empty"), scene, scenes.default_settings(WIDTH=32, HEIGHT=24))
    for d in (dark, empty):
        hs = pt.load_scene_directory(d)
        n = hs.settings.width * hs.settings.height
        job = pt.make_job(0, n, 2, 0, 1)
        got, want = pt.DeviceScene(hs).trace_samples(job), ob.OracleScene(hs).trace_samples(job)
        for g, w in zip(got, want):
            np.testing.assert_allclose(g, w, rtol=1e-9, atol=1e-12)
    assert np.allclose(got[0], [0.25, 0.5, 0.75])
    # jobs of zero pixels or zero samples change nothing; a one-pixel job works
    hs = pt.load_scene_directory(dark)
    dev, film = pt.DeviceScene(hs), pt.Film(32, 24)
    dev.render(film, pt.make_job(5, 1, 3, 0, 1))
    dev.render(film, pt.make_job(0, 0, 3, 0, 1))
    dev.render(film, pt.make_job(0, 32 * 24, 0, 0, 1))
    counters = film.download()[3]
    assert counters.sum() == 3 and counters.reshape(-1)[5] == 3
    # out-of-range jobs are refused, not clamped
    with pytest.raises(pt.PtkError):
        dev.render(film, pt.make_job(32 * 24 - 1, 2, 1, 0, 1))


# traverse_lanes = 1 (default): k_traverse walks the hierarchies of the queued rays with one lane per ray;
# 4: k_traverse_quad, four lanes per ray.  Exact ties between triangles aside (none in these scenes), the closest hit is the same.
@pytest.mark.parametrize("name", ["mesh_gold", "mesh_rough_glass", "mesh_glass", "mesh_textured", "lamp", "thicket", "room"])
def test_both_traversal_kernels_give_the_same_results(scene_dirs, name):
    hs = pt.load_scene_directory(scene_dirs[name])
    n = hs.settings.width * hs.settings.height
    job = pt.make_job(0, n, 5, 1, 41)
    pt.set_option("tail_records", 0)  # every bounce through the deferral and the traversal kernel
    try:
        dev = pt.DeviceScene(hs)
        one = dev.trace_samples(job)
        assert dev.stats().traversed_rays > 0
        pt.set_option("traverse_lanes", 4)
        four = dev.trace_samples(job)
    finally:
        pt.set_option("traverse_lanes", 1)
        pt.set_option("tail_records", 1048576)
    for a, b in zip(four, one):
        assert np.array_equal(a, b)
    assert dev.stats().stack_overflows == 0


@pytest.mark.parametrize("name", ["example", "mesh_gold", "cloud"])
def test_interleaved_band_jobs_reproduce_the_frame(scene_dirs, name):
    """The tile half of the multi-GPU split on one GPU: the interleaved row bands of each of three would-be ranks,
    rendered as one banded job per rank (ptk_render_bands), give the film of the single whole-frame job bit for bit,
    also when the path pool is small enough to cut a rank's job into batches in the middle of a band."""
    from pathtracing_amd.distributed import band_job
    hs = pt.load_scene_directory(scene_dirs[name])
    W, H = hs.settings.width, hs.settings.height
    dev = pt.DeviceScene(hs)
    whole = pt.Film(W, H)
    dev.render(whole, pt.make_job(0, W * H, 5, 2, 99))
    want = whole.download()
    for in_flight in (0, 1500):
        banded_scene = pt.DeviceScene(hs, max_paths_in_flight=in_flight)
        film = pt.Film(W, H)
        for rank in range(3):
            start, n, band_pixels, stride = band_job(rank, 3, W, H, band_rows=4)
            banded_scene.render_bands(film, pt.make_job(start, n, 5, 2, 99), band_pixels, stride)
        got = film.download()
        for a, b in zip(want, got):
            assert np.array_equal(a, b), in_flight
    with pytest.raises(pt.PtkError):  # the last band would leave the image
        dev.render_bands(whole, pt.make_job(W * (H - 2), 4 * W, 1, 0, 0), 4 * W, 12 * W)


# scene, builder, least reference spp, and the tolerances: relative RMSE and mean over the 98 % of pixels that differ
# least, relative mean absolute error over all pixels
CONVERGED = [("example", "cornell_example", 32768, 0.05, 0.005, 0.08), ("cloud", "cloud_box", 32768, 0.05, 0.005, 0.08),
             ("mesh_gold", "mesh_box", 16384, 0.25, 0.03, 0.35)]


@pytest.mark.parametrize("name,builder,least_spp,rmse_tol,mean_tol,mae_tol", CONVERGED)
def test_converged_image_matches_the_reference(tmp_path, name, builder, least_spp, rmse_tol, mean_tol, mae_tol):
    """The GPU's counter-based random streams against the reference's own generators, statistically: a 65,536-spp GPU
    render of a 48x32 frame against the reference's mean image at 65,536 spp (example) / 32,768 spp (scattering cloud) /
    16,384 spp (gold mesh of 4096 triangles), tests/golden/converged_*.npz.  The estimator has heavy tails (caustic paths
    through the mirror and glass balls put rare, very bright samples into a few pixels of either image, and quadrupling
    the sample count of either side does not tame the plain RMSE), so the statistics leave out the 2 % of the pixels
    that differ most:
        per-channel relative RMSE over the other 98 %   <= 0.05   (measured: 0.033 example, 0.019 cloud)
        per-channel mean over the same pixels           within 0.5 %   (measured: 0.08 %, 0.24 %)
        per-channel relative mean absolute error, all pixels   <= 0.08   (measured: 0.055, 0.018)
    The gold mesh is the hardest of the three: light reaches most pixels over glossy bounces between the mesh and the
    red wall, the plain relative RMSE of its red channel is 2.5 at these sample counts, and the reference image has a
    quarter of the samples.  Its tolerances are 0.25 / 3 % / 0.35 (measured, worst channel: 0.157 / 1.3 % / 0.25;
    green and blue 0.054 / 0.5 % / 0.08 and 0.027 / 0.3 % / 0.02)."""
    import os
    from pathtracing_amd import scenes
    ref = np.load(os.path.join(os.path.dirname(__file__), "golden", f"converged_{name}.npz"))
    want = ref["image"].astype(np.float64)
    assert int(ref["spp"]) >= least_spp
    d = str(tmp_path / name)
    getattr(scenes, builder)(d, WIDTH=48, HEIGHT=32)
    hs = pt.load_scene_directory(d)
    dev, film = pt.DeviceScene(hs), pt.Film(48, 32)
    for k in range(8):
        dev.render(film, pt.make_job(0, 48 * 32, 8192, k * 8192, 5))
    got = film.download()[0]
    n_outliers = int(0.02 * 48 * 32)
    for c in range(3):
        g, w = got[..., c].ravel(), want[..., c].ravel()
        keep = np.argsort((g - w) ** 2)[:-n_outliers]
        assert np.sqrt(np.mean((g[keep] - w[keep]) ** 2)) / w.mean() <= rmse_tol, c
        assert abs(g[keep].mean() - w[keep].mean()) / w[keep].mean() <= mean_tol, c
        assert np.mean(np.abs(g - w)) / w.mean() <= mae_tol, c


def test_record_staging_through_lds_delivers_every_record():
    """k_bounce prefetches its records with asynchronous loads straight into LDS and reads them back behind
    hand-written wait counts (request_record / staged_record in path_kernels.h).  The library's self-test pushes
    records through that very pair, with other loads in flight and lanes switched off; none may differ."""
    import ctypes as C
    bad = C.c_int64(-1)
    for n in (1, 255, 4096, 1 << 20):
        assert pt.device_lib().ptk_selftest_record_staging(n, C.byref(bad)) == 0, pt.device_lib().ptk_last_error()
        assert bad.value == 0, (n, bad.value)


def test_tuning_knobs_reject_what_they_do_not_know():
    for name, value in (("batch_lanes", 5), ("batch_lanes", -1), ("launch_ahead", 8), ("launch_ahead", -1), ("tail_records", -1),
                        ("tail_records", (1 << 24) + 1), ("traverse_lanes", 2), ("gpu_bvh_build", 2), ("no_such_knob", 1)):
        with pytest.raises(RuntimeError):
            pt.set_option(name, value)
    for name, value in (("batch_lanes", 0), ("launch_ahead", 7), ("tail_records", 1 << 20), ("traverse_lanes", 1), ("gpu_bvh_build", 0)):
        pt.set_option(name, value)  # the defaults


# Two batches in flight, each on a stream and a host thread of its own, at a size the oracle can follow: a job of 2^21
# paths in a scene with a mesh, against the same job traced one batch at a time and against the oracle on a run of it.
def test_two_batches_in_flight_give_the_same_samples(scene_dirs, ob):
    hs = pt.load_scene_directory(scene_dirs["mesh_rough_glass"])
    n = hs.settings.width * hs.settings.height
    spp = (1 << 21) // n + 1
    job = pt.make_job(0, n, spp, 0, 5)
    pt.set_option("batch_lanes", 2)
    try:
        dev = pt.DeviceScene(hs)
        two = dev.trace_samples(job)
        two_stats = dev.stats()
        pt.set_option("batch_lanes", 1)
        dev = pt.DeviceScene(hs)
        one = dev.trace_samples(job)
        one_stats = dev.stats()
    finally:
        pt.set_option("batch_lanes", 0)
    for a, b in zip(two, one):
        assert np.array_equal(a, b)
    for field in ("paths", "vertices", "extension_rays", "shadow_rays", "traversed_rays", "nodes_visited", "triangles_tested"):
        assert getattr(two_stats, field) == getattr(one_stats, field), field
    assert two_stats.paths == n * spp
    small = pt.make_job(100, 64, 4, 0, 5)
    got, want = pt.DeviceScene(hs).trace_samples(small), ob.OracleScene(hs).trace_samples(small)
    assert np.isclose(got[0], want[0], rtol=1e-9, atol=1e-12).all(axis=-1).mean() >= 0.99


def test_film_merge_and_scene_clone(scene_dirs):
    """The combine step of several GPUs rendering one image, on one device: a cloned scene (ptk_scene_clone: tables copied
    device to device, nothing built again) renders the same bits as its source; films of disjoint row bands fold
    (ptk_film_merge) into the whole frame bit for bit; films of sample slices fold into what increment_average's
    arithmetic (app/pathtracer.cpp:232-235) gives on the host, bit for bit."""
    hs = pt.load_scene_directory(scene_dirs["mesh_gold"])
    W, H = hs.settings.width, hs.settings.height
    n = W * H
    dev = pt.DeviceScene(hs)
    twin = dev.clone()
    whole = pt.Film(W, H)
    dev.render(whole, pt.make_job(0, n, 6, 0, 3))
    want = whole.download()
    # row bands of 8 rows, dealt to the two scenes in turn, each into a film of its own
    parts = [pt.Film(W, H), pt.Film(W, H)]
    for k, scene in enumerate((dev, twin)):
        start, pixels, band_pixels, stride = band_job(k, 2, W, H, 8)
        scene.render_bands(parts[k], pt.make_job(start, pixels, 6, 0, 3), band_pixels, stride)
    merged = pt.Film(W, H)
    merged.merge(parts[0])
    merged.merge(parts[1])
    for a, b in zip(merged.download(), want):
        assert np.array_equal(a, b)
    # sample slices: the source renders samples 0..3, the clone 4..5 of every pixel
    slices = [pt.Film(W, H), pt.Film(W, H)]
    dev.render(slices[0], pt.make_job(0, n, 4, 0, 3))
    twin.render(slices[1], pt.make_job(0, n, 2, 4, 3))
    a, b = slices[0].download(), slices[1].download()
    slices[0].merge(slices[1])
    got = slices[0].download()
    for k in range(3):
        assert np.array_equal(got[k], (a[k] * 4.0 + b[k] * 2.0) / 6.0)
        np.testing.assert_allclose(got[k], want[k], rtol=1e-12, atol=1e-15)
    assert (got[3] == 6).all()
    assert np.array_equal(b[0], slices[1].download()[0])  # the film that was folded in is left as it was
    with pytest.raises(pt.PtkError):
        merged.merge(pt.Film(W + 1, H))
