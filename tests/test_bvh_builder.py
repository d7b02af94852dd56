"""The host BVH builder (pathtracing_amd/csrc/host/bvh_builder.cpp, which replaces the reference's BVH build,
src/bvh.cpp:91-165) checked structurally on random triangles by a small C++ program built with g++ and the
address / undefined-behaviour sanitizers: child references in range, every triangle in exactly one leaf, every node
reachable, child boxes containing their subtrees."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("bvh") / "bvh_check")
    subprocess.run(
        ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-static-libasan", "-static-libubsan", "-I" + os.path.join(ROOT, "include"),
         "-I" + os.path.join(ROOT, "pathtracing_amd", "csrc"), os.path.join(ROOT, "tests", "cpp", "bvh_check.cpp"),
         os.path.join(ROOT, "pathtracing_amd", "csrc", "host", "bvh_builder.cpp"), "-o", exe, "-lpthread"],
        check=True)
    return exe


@pytest.mark.parametrize("n,max_leaf,seed", [(0, 2, 1), (1, 2, 1), (2, 2, 1), (5, 2, 2), (100, 2, 3), (5000, 2, 4),
                                             (20000, 4, 5), (3, 8, 6), (777, 1, 7)])
def test_builder_output_is_a_valid_hierarchy(checker, n, max_leaf, seed):
    out = subprocess.run([checker, str(n), str(max_leaf), str(seed)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout + out.stderr
