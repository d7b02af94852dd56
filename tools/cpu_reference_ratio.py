#!/usr/bin/env python3
"""Throughput of the CPU restatement (oracle/) against the reference itself (oracle/_ref, built from the
reference's sources), on the same scene, in the build container.  Prints Mpaths/s for both at 1 and N threads.

usage: tools/cpu_reference_ratio.py [width height spp threads]
"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import pathtracing_amd as pt  # noqa: E402
from oracle import binding as ob  # noqa: E402
from pathtracing_amd import scenes  # noqa: E402


def main():
    W, H, spp, threads = (int(a) for a in (sys.argv[1:5] if len(sys.argv) >= 5 else (256, 256, 4, 7)))
    d = scenes.cornell_example(tempfile.mkdtemp(prefix="ptk_ratio_"), WIDTH=W, HEIGHT=H, samples_per_pixel=spp,
                               samples_per_iteration=spp)
    hs = pt.load_scene_directory(d)
    job = pt.make_job(0, W * H, spp, 0, 1)
    buffers = lambda: (np.zeros((H, W, 3)), np.zeros((H, W, 3)), np.zeros((H, W, 3)), np.zeros((H, W), dtype=np.uint64))  # noqa: E731
    for name, scene in (("reference", ob.RefScene(hs)), ("restatement", ob.OracleScene(hs))):
        for t in (1, threads):
            t0 = time.perf_counter()
            if name == "reference":
                scene.render(threads=t)  # the reference's own thread_job loop over the whole settings
            else:
                scene.render(job, *buffers(), threads=t)
            dt = time.perf_counter() - t0
            print(f"{name:12s} threads={t:2d}  {W * H * spp / dt / 1e6:8.3f} Mpaths/s  ({dt:.1f} s)")


if __name__ == "__main__":
    main()
