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


# Meshes that are nothing like a tessellated surface.  The walk on the GPU has 64 stack entries; the builder reports
# what its tree needs (the checker recomputes it from the tree's shape) and builds again with a shallower tree when the
# surface-area heuristic's would need more: "sliver" needs about 100 entries without that.
@pytest.mark.parametrize("n,max_leaf,seed,shape,limit", [(6000, 2, 3, "sliver", 64), (50000, 1, 3, "sliver", 64),
                                                         (50000, 1, 3, "sliver", 30), (60000, 2, 5, "clustered", 64),
                                                         (60000, 2, 5, "clustered", 20)])
def test_stack_need_is_bounded_on_lopsided_meshes(checker, n, max_leaf, seed, shape, limit):
    out = subprocess.run([checker, str(n), str(max_leaf), str(seed), shape, str(limit)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout + out.stderr
    stack = int(out.stdout.split("stack=")[1].split()[0])
    assert stack <= 64
    if shape == "sliver" and limit == 64:
        assert "sah_levels=56" not in out.stdout  # the fallback was exercised
