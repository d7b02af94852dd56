"""The host code under sanitizers (tools/sanitize_host.sh): the multi-threaded hierarchy build under the thread
sanitizer on uniform and lopsided meshes, and the scene loader under the address and undefined-behaviour sanitizers
on scene directories mutated by tools/fuzz_loader.py.  CPU builds with g++; no GPU involved.  (The reference has no
sanitizer run of any kind; its known races are listed where the scheduler is described, app/pathtracer.cpp:277-289.)"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "tools", "sanitize_host.sh")


def test_multi_threaded_hierarchy_build_and_obj_parse_are_race_free_under_tsan():
    r = subprocess.run(["bash", SCRIPT, "thread"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "thread sanitizer: no report" in r.stdout and r.stdout.count("\nok ") + r.stdout.startswith("ok ") == 5
    assert "loaded 8 9216" in r.stdout  # the mesh scene parsed in eight parts: 8 objects, 9216 triangles


def test_fuzzed_scene_directories_load_cleanly_under_asan_and_ubsan():
    r = subprocess.run(["bash", SCRIPT, "address", "0", "40"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-6000:] + r.stderr[-3000:]
    assert "address and undefined-behaviour sanitizers: no report" in r.stdout
