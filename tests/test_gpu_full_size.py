"""GPU checks at the benchmark's own size (1920x1080, 8 samples per pixel per job): properties that do not
need the oracle to render the whole frame (partition and batch invariance, determinism, counters), plus
oracle comparisons on pixel runs scattered over the frame and on the full-frame denoise."""
import numpy as np
import pytest

import pathtracing_amd as pt
from pathtracing_amd import scenes

pytestmark = pytest.mark.gpu
W, H, SPP = 1920, 1080, 8
N = W * H


@pytest.fixture(scope="module", autouse=True)
def gpu():
    pt.init(0)


@pytest.fixture(scope="module")
def example(tmp_path_factory):
    d = scenes.cornell_example(str(tmp_path_factory.mktemp("full_example")), WIDTH=W, HEIGHT=H)
    return pt.load_scene_directory(d)


@pytest.fixture(scope="module")
def mesh(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("full_mesh"))
    scenes.mesh_box(d, n_lon=512, n_lat=257, material="gold", smooth=True, WIDTH=W, HEIGHT=H)  # 262,144 triangles
    return pt.load_scene_directory(d)


@pytest.fixture(scope="module")
def cloud(tmp_path_factory):
    d = scenes.cloud_box(str(tmp_path_factory.mktemp("full_cloud")), WIDTH=W, HEIGHT=H)  # scattering medium + coloured glass
    return pt.load_scene_directory(d)


@pytest.fixture(scope="module")
def room(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("full_room"))
    scenes.living_room(d, mesh_lon=128, mesh_lat=65, texture_size=512, WIDTH=W, HEIGHT=H)  # 23 objects, 5 textured meshes
    return pt.load_scene_directory(d)


def render(host_scene, jobs, **kwargs):
    dev = pt.DeviceScene(host_scene, **kwargs)
    film = pt.Film(W, H)
    for job in jobs:
        dev.render(film, job)
    return film.download(), dev.stats()


@pytest.mark.parametrize("which", ["example", "mesh", "cloud", "room"])
def test_frame_is_the_same_however_it_is_cut(which, request):
    hs = request.getfixturevalue(which)
    whole, stats = render(hs, [pt.make_job(0, N, SPP, 0, 11)])
    image, position, normal, counters = whole
    assert np.isfinite(image).all() and (image >= 0).all() and (counters == SPP).all()
    assert stats.paths == N * SPP and stats.vertices >= stats.paths * 0.9
    # again: bit-identical (no atomics on floating-point data, no dependence on scheduling)
    again, _ = render(hs, [pt.make_job(0, N, SPP, 0, 11)])
    assert all(np.array_equal(a, b) for a, b in zip(whole, again))
    # pixel ranges and small batches: bit-identical
    cut = N // 3 + 17
    pieces, _ = render(hs, [pt.make_job(0, cut, SPP, 0, 11), pt.make_job(cut, N - cut, SPP, 0, 11)], max_paths_in_flight=3_000_000)
    assert all(np.array_equal(a, b) for a, b in zip(whole, pieces))
    # every bounce through the class sort and k_bounce (0), or the whole frame through the persistent k_tail (2^24):
    # bit-identical to the default, which switches from the one to the other part-way
    for tail_records in (0, 1 << 24):
        pt.set_option("tail_records", tail_records)
        try:
            other, _ = render(hs, [pt.make_job(0, N, SPP, 0, 11)])
        finally:
            pt.set_option("tail_records", 1 << 20)
        assert all(np.array_equal(a, b) for a, b in zip(whole, other)), tail_records
    # one batch at a time, or two, three, four (each on a stream and a host thread of its own), the host far ahead of the device's record
    # counts or not at all: bit-identical
    for option, value, default in (("batch_lanes", 1, 0), ("batch_lanes", 2, 0), ("batch_lanes", 3, 0), ("batch_lanes", 4, 0), ("launch_ahead", 0, 7)):
        pt.set_option(option, value)
        try:
            other, other_stats = render(hs, [pt.make_job(0, N, SPP, 0, 11)])
        finally:
            pt.set_option(option, default)
        assert all(np.array_equal(a, b) for a, b in zip(whole, other)), (option, value)
        assert (other_stats.paths, other_stats.vertices) == (stats.paths, stats.vertices), (option, value)
    # sample slices folded one after the other (what two GPUs would add up): equal up to rounding of the mean
    slices, _ = render(hs, [pt.make_job(0, N, SPP // 2, 0, 11), pt.make_job(0, N, SPP - SPP // 2, SPP // 2, 11)])
    np.testing.assert_allclose(slices[0], image, rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(slices[1], position, rtol=1e-12, atol=1e-15)
    assert np.array_equal(slices[3], counters)


@pytest.mark.parametrize("which", ["example", "mesh", "cloud", "room"])
def test_scattered_pixel_runs_match_the_oracle(which, request):
    from oracle import binding as ob
    hs = request.getfixturevalue(which)
    dev, osc = pt.DeviceScene(hs), ob.OracleScene(hs)
    rng = np.random.default_rng(5)
    close, total = 0, 0
    sums = np.zeros((2, 3))
    for start in rng.integers(0, N - 256, size=12):
        job = pt.make_job(int(start), 256, SPP, 3, 77)
        got, want = dev.trace_samples(job), osc.trace_samples(job)
        g, w = got[0].reshape(-1, 3), want[0].reshape(-1, 3)
        close += np.isclose(g, w, rtol=1e-9, atol=1e-12).all(axis=1).sum()
        total += len(g)
        sums += [g.sum(axis=0), w.sum(axis=0)]
        np.testing.assert_allclose(got[1], want[1], rtol=1e-9, atol=1e-9)  # first-surface positions
    assert close / total >= 0.999, close / total
    np.testing.assert_allclose(sums[0], sums[1], rtol=5e-3)


def test_full_frame_denoise_matches_the_oracle(example):
    from oracle import binding as ob
    (image, position, normal, _), _ = render(example, [pt.make_job(0, N, 2, 0, 3)])
    tasks = pt.make_denoise_tasks([{"mode": "median", "kernel_size": 3, "threshold": 0.5},
                                   {"mode": "atrous", "iterations": 4, "sigma_rt": 2, "sigma_x": 2, "sigma_n": 2}])
    got = pt.denoise(tasks, image, position, normal)
    want = ob.oracle_denoise(tasks, image, position, normal)
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-13)
    # a constant image stays constant: the tap weights are normalised
    flat = np.full_like(image, 0.25)
    np.testing.assert_allclose(pt.denoise(tasks, flat, position, normal), flat, rtol=1e-12)
