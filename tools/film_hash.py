"""Prints hashes of the film a device-library build renders for a fixed set of scenes and jobs, one line per scene:
first with the default switch to the tail kernel (a frame this small is traced by k_generate and k_tail alone), then
with tail_records = 0, which sends the same job through the class sort and k_bounce (with k_traverse and
k_bounce_resume in scenes with meshes) as a full-size frame is.  Two builds whose lines agree render the same films
bit for bit: the check behind every change to the kernels that is meant to cost less without changing a result
(select the build with PTK_DEVICE_LIBRARY, as for bench.py).
usage: python tools/film_hash.py [first_random_seed last_random_seed]"""
import hashlib
import os
import shutil
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracing_amd as pt
from pathtracing_amd import scenes

W, H, SPP = 640, 360, 4


def film_hash(host_scene, seed):
    dev = pt.DeviceScene(host_scene)
    film = pt.Film(W, H)
    dev.render(film, pt.make_job(0, W * H, SPP, 0, seed))
    digest = hashlib.sha256()
    for buffer in film.download():
        digest.update(np.ascontiguousarray(buffer).tobytes())
    return digest.hexdigest()[:16]


def main():
    pt.init(0)
    first, last = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1, 12)
    builders = [("example", lambda d: scenes.cornell_example(d, WIDTH=W, HEIGHT=H)),
                ("zoo", lambda d: scenes.material_zoo(d, WIDTH=W, HEIGHT=H)),
                ("zoo_circle", lambda d: scenes.material_zoo(d, dof="circle", WIDTH=W, HEIGHT=H)),
                ("cloud", lambda d: scenes.cloud_box(d, WIDTH=W, HEIGHT=H)),
                ("fog", lambda d: scenes.cloud_box(d, fog=True, WIDTH=W, HEIGHT=H)),
                ("mesh_gold", lambda d: scenes.mesh_box(d, WIDTH=W, HEIGHT=H)),
                ("mesh_textured", lambda d: scenes.mesh_box(d, textured=True, WIDTH=W, HEIGHT=H)),
                ("thicket", lambda d: scenes.thicket_box(d, WIDTH=W, HEIGHT=H)),
                ("lamp", lambda d: scenes.emissive_mesh_box(d, WIDTH=W, HEIGHT=H)),
                ("room", lambda d: scenes.living_room(d, mesh_lon=32, mesh_lat=17, texture_size=64, WIDTH=W, HEIGHT=H))]
    for seed in range(first, last + 1):
        builders.append((f"random_{seed}", lambda d, seed=seed: scenes.random_box(d, seed, extras=bool(seed % 2), WIDTH=W, HEIGHT=H)))
    for name, build in builders:
        directory = tempfile.mkdtemp(prefix="ptk_hash_")
        try:
            build(directory) # (some builders return the directory, some a triangle count)
            host_scene = pt.load_scene_directory(directory)
        finally:
            shutil.rmtree(directory, ignore_errors=True)
        pt.set_option("tail_records", 1048576)
        by_tail = film_hash(host_scene, 11)
        pt.set_option("tail_records", 0)
        by_bounces = film_hash(host_scene, 11)
        print(name, by_tail, by_bounces, flush=True)


if __name__ == "__main__":
    main()
