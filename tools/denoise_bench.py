#!/usr/bin/env python3
"""Times the denoise kernels on a 1920x1080 frame of the scattering-medium scene (benchmark configuration 5): the
reference's shipped pipeline (median 3x3 + a-trous x4), then single tasks.  Meant to run under
`rocprofv3 --kernel-trace --stats` for per-kernel times.  usage: tools/denoise_bench.py [reps]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracing_amd as pt  # noqa: E402
from pathtracing_amd import scenes  # noqa: E402

W, H = 1920, 1080


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    pt.init(0)
    d = scenes.cloud_box(tempfile.mkdtemp(prefix="ptk_dn_"), WIDTH=W, HEIGHT=H)
    hs = pt.load_scene_directory(d)
    dev, film, scratch = pt.DeviceScene(hs), pt.Film(W, H), pt.Film(W, H)
    dev.render(film, pt.make_job(0, W * H, 8, 0, 1))
    shipped = [dict(mode=t["mode"], **t["parameters"]) for t in scenes.DEFAULT_DENOISING["pipeline"]]
    cases = {"shipped pipeline (median 3 + a-trous x4)": shipped,
             "median 3": [dict(mode="median", kernel_size=3, threshold=0.5)],
             "median 5": [dict(mode="median", kernel_size=5, threshold=0.5)],
             "median 9": [dict(mode="median", kernel_size=9, threshold=0.5)],
             "median 15": [dict(mode="median", kernel_size=15, threshold=0.5)],
             "a-trous x1": [dict(mode="atrous", iterations=1, sigma_rt=2, sigma_x=2, sigma_n=2)],
             "a-trous x4": [dict(mode="atrous", iterations=4, sigma_rt=2, sigma_x=2, sigma_n=2)]}
    for name, pipeline in cases.items():
        tasks = pt.make_denoise_tasks(pipeline)
        pt.denoise_film(tasks, film, scratch)
        pt.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            pt.denoise_film(tasks, film, scratch)
        pt.sync()
        ms = (time.perf_counter() - t0) / reps * 1e3
        print(f"{name}: {ms:.3f} ms per 1920x1080 frame ({W * H / ms / 1e3:.0f} Mpixel/s)")


if __name__ == "__main__":
    main()
