"""The hierarchy build on the GPU (csrc/device/bvh_build.hip: Morton-ordered linear BVH collapsed to 4-wide nodes; it
replaces, like the host builder, the reference's BVH::Node::Node, src/bvh.cpp:129-165) against the host build: any valid
hierarchy gives the same closest hit, so primary hits (object, face, distance, position, normal) and per-sample radiance
must be the same bit for bit (no two triangles of these meshes tie exactly), and the oracle must agree as it does with
the host-built tree."""
import numpy as np
import pytest

import pathtracing_amd as pt
from pathtracing_amd import scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def gpu():
    pt.init(0)
    yield
    pt.set_option("gpu_bvh_build", 0)


def both_builds(host_scene):
    pt.set_option("gpu_bvh_build", 0)
    on_host = pt.DeviceScene(host_scene)
    pt.set_option("gpu_bvh_build", 1)
    try:
        on_gpu = pt.DeviceScene(host_scene)
    finally:
        pt.set_option("gpu_bvh_build", 0)
    return on_host, on_gpu


@pytest.mark.parametrize("kwargs", [dict(material="gold"), dict(material="glass", n_lon=40, n_lat=27), dict(textured=True, smooth=False)],
                         ids=["gold", "glass", "textured"])
@pytest.mark.parametrize("tail_records", [1048576, 0])
def test_gpu_built_hierarchy_gives_the_same_hits(tmp_path, kwargs, tail_records):
    d = str(tmp_path / "mesh")
    n_tri = scenes.mesh_box(d, WIDTH=96, HEIGHT=64, **kwargs)
    hs = pt.load_scene_directory(d)
    on_host, on_gpu = both_builds(hs)
    info = on_gpu.build_info()
    assert info["meshes_built_on_gpu"] == 1 and info["meshes_built_on_host"] == 0 and info["bvh_nodes"] > n_tri // 8
    assert on_host.build_info()["meshes_built_on_gpu"] == 0
    n = 96 * 64
    primary = pt.make_job(0, n, 1, 0, 4)
    for a, b in zip(on_host.trace_primary(primary), on_gpu.trace_primary(primary)):
        assert np.array_equal(a, b)
    job = pt.make_job(0, n, 4, 1, 19)
    pt.set_option("tail_records", tail_records)
    try:
        want, got = on_host.trace_samples(job), on_gpu.trace_samples(job)
    finally:
        pt.set_option("tail_records", 1048576)
    for a, b in zip(want, got):
        assert np.array_equal(a, b)
    assert on_gpu.stats().stack_overflows == 0


def test_small_meshes_and_lopsided_ones_fall_back_to_the_host_build(scene_dirs):
    pt.set_option("gpu_bvh_build", 1)
    try:
        lamp = pt.DeviceScene(pt.load_scene_directory(scene_dirs["lamp"]))  # an emissive mesh of a few hundred triangles
        thicket = pt.DeviceScene(pt.load_scene_directory(scene_dirs["thicket"]))  # 6000 triangles, nested and clustered
    finally:
        pt.set_option("gpu_bvh_build", 0)
    assert lamp.build_info()["meshes_built_on_gpu"] == 0 and lamp.build_info()["meshes_built_on_host"] >= 1
    info = thicket.build_info()
    assert info["meshes_built_on_gpu"] + info["meshes_built_on_host"] == 1
    hs = pt.load_scene_directory(scene_dirs["thicket"])
    job = pt.make_job(0, hs.settings.width * hs.settings.height, 1, 0, 4)
    for a, b in zip(thicket.trace_primary(job), pt.DeviceScene(hs).trace_primary(job)):
        assert np.array_equal(a, b)


def test_a_build_the_device_has_no_memory_for_falls_back_to_the_host(scene_dirs):
    """With gpu_bvh_build on, a hierarchy build that runs out of device memory (forced here by capping what the build
    may allocate) must not refuse a scene that loads with the option off: the mesh is built on the host, no error is
    left pending, and the scene traces like the host-built one."""
    hs = pt.load_scene_directory(scene_dirs["thicket"])  # 6000 triangles: about 2 MB of build memory
    pt.set_option("gpu_bvh_build", 1)
    pt.set_option("gpu_bvh_memory_limit", 65536)
    try:
        starved = pt.DeviceScene(hs)
    finally:
        pt.set_option("gpu_bvh_memory_limit", 0)
        pt.set_option("gpu_bvh_build", 0)
    info = starved.build_info()
    assert info["meshes_built_on_gpu"] == 0 and info["meshes_built_on_host"] == 1 and info["bvh_build_wall_ms"] > 0
    pt.sync()  # (a pending HIP error would surface here)
    job = pt.make_job(0, hs.settings.width * hs.settings.height, 1, 0, 4)
    for a, b in zip(starved.trace_primary(job), pt.DeviceScene(hs).trace_primary(job)):
        assert np.array_equal(a, b)
    with pytest.raises(Exception):
        pt.set_option("gpu_bvh_memory_limit", -1)


def test_million_triangles_build_time_and_oracle_parity(tmp_path):
    from oracle import binding as ob
    d = str(tmp_path / "big")
    assert scenes.mesh_box(d, n_lon=1024, n_lat=513, material="rough_glass", WIDTH=640, HEIGHT=360) == 1048576
    hs = pt.load_scene_directory(d)
    on_host, on_gpu = both_builds(hs)
    host_ms, gpu_ms = on_host.build_info()["bvh_build_ms"], on_gpu.build_info()["bvh_build_ms"]
    gpu_wall_ms = on_gpu.build_info()["bvh_build_wall_ms"]
    print(f"1,048,576 triangles: host build {host_ms:.0f} ms, GPU build {gpu_ms:.1f} ms of device time, {gpu_wall_ms:.0f} ms by the host's clock")
    assert gpu_wall_ms >= gpu_ms
    assert on_gpu.build_info()["meshes_built_on_gpu"] == 1 and 0 < gpu_ms < host_ms / 10
    n = 640 * 360
    primary = pt.make_job(0, n, 1, 0, 4)
    for a, b in zip(on_host.trace_primary(primary), on_gpu.trace_primary(primary)):
        assert np.array_equal(a, b)
    job = pt.make_job(180 * 640 + 200, 256, 4, 0, 6)
    got, want = on_gpu.trace_samples(job), ob.OracleScene(hs).trace_samples(job)
    ok = np.isclose(got[0].reshape(-1, 3), want[0].reshape(-1, 3), rtol=1e-9, atol=1e-12).all(axis=1)
    assert ok.mean() >= 0.995
    film_host, film_gpu = pt.Film(640, 360), pt.Film(640, 360)
    pt.set_option("tail_records", 0)  # every bounce through k_traverse, which counts the nodes it visits
    try:
        on_host.render(film_host, pt.make_job(0, n, 8, 0, 2))
        on_gpu.render(film_gpu, pt.make_job(0, n, 8, 0, 2))
    finally:
        pt.set_option("tail_records", 1048576)
    assert np.array_equal(film_host.download()[0], film_gpu.download()[0])
    sh, sg = on_host.stats(), on_gpu.stats()
    print(f"nodes visited per traversed ray: host-built {sh.nodes_visited / max(sh.traversed_rays, 1):.1f}, "
          f"GPU-built {sg.nodes_visited / max(sg.traversed_rays, 1):.1f}")
    assert sg.stack_overflows == 0
