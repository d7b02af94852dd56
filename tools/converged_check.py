#!/usr/bin/env python3
"""Prints how far a high-sample-count GPU render is from the reference's converged mean images
(tests/golden/converged_*.npz): per-channel relative RMSE, relative mean absolute error and relative mean difference.
usage: tools/converged_check.py [spp, default 65536]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracing_amd as pt  # noqa: E402
from pathtracing_amd import scenes  # noqa: E402


def trimmed_mean_difference(got, want):
    keep = np.argsort(((got - want) ** 2).ravel())[:-31]
    return (got.ravel()[keep].mean() - want.ravel()[keep].mean()) / want.ravel()[keep].mean()


def main():
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    pt.init(0)
    for name, builder in (("example", scenes.cornell_example), ("cloud", scenes.cloud_box), ("mesh_gold", scenes.mesh_box)):
        ref = np.load(os.path.join(ROOT, "tests", "golden", f"converged_{name}.npz"))
        want = ref["image"].astype(np.float64)
        d = tempfile.mkdtemp(prefix="ptk_conv_")
        builder(d, WIDTH=48, HEIGHT=32)  # (mesh_box returns its triangle count, not the directory)
        hs = pt.load_scene_directory(d)
        dev, film = pt.DeviceScene(hs), pt.Film(48, 32)
        for k in range(spp // 8192):
            dev.render(film, pt.make_job(0, 48 * 32, 8192, k * 8192, 5))
        got = film.download()[0]
        for c in range(3):
            mean = want[..., c].mean()
            print(name, "rgb"[c], "reference spp", int(ref["spp"]), "gpu spp", spp,
                  "rRMSE %.4f" % (np.sqrt(np.mean((got[..., c] - want[..., c]) ** 2)) / mean),
                  "trimmed rRMSE %.4f" % (np.sqrt(np.mean(np.sort(((got[..., c] - want[..., c]) ** 2).ravel())[:-31])) / mean),
                  "rMAE %.4f" % (np.mean(np.abs(got[..., c] - want[..., c])) / mean),
                  "mean diff %.5f" % ((got[..., c].mean() - mean) / mean),
                  "mean diff without the 2%% of pixels that differ most %.5f" % trimmed_mean_difference(got[..., c], want[..., c]))


if __name__ == "__main__":
    main()
