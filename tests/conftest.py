import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def scene_root(tmp_path_factory):
    return tmp_path_factory.mktemp("scenes")


@pytest.fixture(scope="session")
def scene_dirs(scene_root):
    """Small generated scene directories covering every feature of the hot path."""
    from pathtracing_amd import scenes
    root = str(scene_root)
    dirs = {}
    dirs["example"] = scenes.cornell_example(os.path.join(root, "example"), WIDTH=64, HEIGHT=48)
    dirs["zoo"] = scenes.material_zoo(os.path.join(root, "zoo"), WIDTH=64, HEIGHT=48)
    dirs["zoo_circle"] = scenes.material_zoo(os.path.join(root, "zoo_circle"), dof="circle", WIDTH=48, HEIGHT=32)
    dirs["zoo_square"] = scenes.material_zoo(os.path.join(root, "zoo_square"), dof="square", WIDTH=48, HEIGHT=32)
    for material in ("gold", "rough_glass", "glass"):
        d = os.path.join(root, "mesh_" + material)
        scenes.mesh_box(d, material=material, WIDTH=64, HEIGHT=48)
        dirs["mesh_" + material] = d
    d = os.path.join(root, "mesh_textured")
    scenes.mesh_box(d, textured=True, smooth=False, WIDTH=64, HEIGHT=48)
    dirs["mesh_textured"] = d
    dirs["cloud"] = scenes.cloud_box(os.path.join(root, "cloud"), WIDTH=64, HEIGHT=48)
    dirs["fog"] = scenes.cloud_box(os.path.join(root, "fog"), fog=True, WIDTH=64, HEIGHT=48)
    dirs["lamp"] = scenes.emissive_mesh_box(os.path.join(root, "lamp"), WIDTH=64, HEIGHT=48)
    dirs["thicket"] = scenes.thicket_box(os.path.join(root, "thicket"), WIDTH=64, HEIGHT=48)
    dirs["room"] = scenes.living_room(os.path.join(root, "room"), mesh_lon=32, mesh_lat=17, texture_size=64, WIDTH=64, HEIGHT=48)
    return dirs
