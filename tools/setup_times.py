#!/usr/bin/env python3
"""Where the set-up time of a large mesh scene goes: writing the OBJ (test fixture only), the host loader (OBJ parse and
triangle set-up), and ptk_scene_create with the hierarchy built on the host and on the GPU (wall time of the whole
call, wall time of the build inside it, device time of the build kernels).
usage: tools/setup_times.py [n_lon n_lat]      default 4096 1281: 10,485,760 triangles"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracing_amd as pt  # noqa: E402
from pathtracing_amd import scenes  # noqa: E402


def main():
    n_lon, n_lat = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 1281)
    pt.init(0)
    d = tempfile.mkdtemp(prefix="ptk_setup_")
    t0 = time.perf_counter()
    n = scenes.mesh_box(d, n_lon=n_lon, n_lat=n_lat, material="gold", smooth=True, WIDTH=1920, HEIGHT=1080)
    t1 = time.perf_counter()
    size = os.path.getsize(os.path.join(d, "models", "bumpy_sphere.obj"))
    print(f"{n} triangles, OBJ of {size / 1e6:.0f} MB written in {t1 - t0:.1f} s (fixture generation, Python)", flush=True)
    hs = pt.load_scene_directory(d)
    t2 = time.perf_counter()
    print(f"host loader (JSON + OBJ parse + triangle set-up): {t2 - t1:.2f} s ({size / 1e6 / (t2 - t1):.0f} MB/s)", flush=True)
    for gpu_build in (0, 1):
        pt.set_option("gpu_bvh_build", gpu_build)
        t3 = time.perf_counter()
        dev = pt.DeviceScene(hs)
        pt.sync()
        t4 = time.perf_counter()
        info = dev.build_info()
        where = "GPU" if info["meshes_built_on_gpu"] else "host"
        print(f"ptk_scene_create, hierarchy on the {where}: {t4 - t3:.2f} s in all; build {info['bvh_build_wall_ms'] / 1e3:.2f} s by the host's clock"
              f" ({info['bvh_build_ms']:.1f} ms {'of device time' if info['meshes_built_on_gpu'] else 'wall'}), {info['bvh_nodes']} nodes;"
              f" the rest (triangle records, tables, upload): {t4 - t3 - info['bvh_build_wall_ms'] / 1e3:.2f} s", flush=True)
        del dev
    pt.set_option("gpu_bvh_build", 0)


if __name__ == "__main__":
    main()
