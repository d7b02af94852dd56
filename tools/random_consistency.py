"""The GPU against itself on generated scenes (scenes.random_box) at a frame large enough for every code path: the
default (class-sorted bounces, then the persistent tail kernel), every bounce through the class sort, the whole
frame through the tail kernel, the general kernels, and the frame cut into batches of 700,001 paths must give
bit-identical films.
usage: python tools/random_consistency.py <first_seed> <last_seed>"""
import os
import shutil
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracing_amd as pt
from pathtracing_amd import scenes

W, H, SPP = 1280, 720, 3
pt.init(0)
first, last = int(sys.argv[1]), int(sys.argv[2])
failures = 0
for seed in range(first, last + 1):
    directory = tempfile.mkdtemp(prefix="ptk_consistency_")
    hs = pt.load_scene_directory(scenes.random_box(directory, seed, extras=bool(seed % 2), WIDTH=W, HEIGHT=H))
    shutil.rmtree(directory, ignore_errors=True)
    films = {}
    for name, tail, specialised, in_flight in (("default", 1 << 20, 1, 0), ("sorted", 0, 1, 0), ("tail", 1 << 24, 1, 0),
                                               ("general", 1 << 20, 0, 0), ("batched", 200_000, 1, 700_001)):
        pt.set_option("tail_records", tail)
        pt.set_option("specialised_kernels", specialised)
        dev = pt.DeviceScene(hs, max_paths_in_flight=in_flight) if in_flight else pt.DeviceScene(hs)
        film = pt.Film(W, H)
        dev.render(film, pt.make_job(0, W * H, SPP, 0, 5 + seed))
        films[name] = film.download()
    pt.set_option("tail_records", 1 << 20)
    pt.set_option("specialised_kernels", 1)
    differing = [name for name in ("sorted", "tail", "general", "batched")
                 if not all(np.array_equal(a, b, equal_nan=True) for a, b in zip(films["default"], films[name]))]
    failures += bool(differing)
    print(seed, "differs: " + ",".join(differing) if differing else "same", flush=True)
print("scenes with differences:", failures)
