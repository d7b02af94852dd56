"""Random sequences of render jobs (pixel ranges, sample counts, sample offsets, small batches; zero-sample jobs, which
must change nothing, on the GPU only) folded into a film on the GPU and by the CPU oracle on generated scenes: counters equal, running means within 1e-9 except
for the rare sample whose path took another branch.  Also through the host-buffer entry point (bit-identical to the
device film).  usage: python tools/random_jobs.py <first_seed> <last_seed>"""
import os
import shutil
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracing_amd as pt
from pathtracing_amd import scenes
from oracle import binding as ob

pt.init(0)
W, H = 40, 28
n = W * H
failures = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2]) + 1):
    rng = np.random.default_rng(3_000_017 + seed)
    directory = tempfile.mkdtemp(prefix="ptk_jobs_")
    hs = pt.load_scene_directory(scenes.random_box(directory, seed, extras=bool(seed % 2), WIDTH=W, HEIGHT=H))
    shutil.rmtree(directory, ignore_errors=True)
    dev = pt.DeviceScene(hs, max_paths_in_flight=int(rng.choice([0, 97, 1000, 5000])))
    film, osc = pt.Film(W, H), ob.OracleScene(hs)
    img, pos, nrm = np.zeros((H, W, 3)), np.zeros((H, W, 3)), np.zeros((H, W, 3))
    cnt = np.zeros((H, W), dtype=np.uint64)
    h_img, h_pos, h_nrm, h_cnt = img.copy(), pos.copy(), nrm.copy(), cnt.copy()
    offset = 0
    for _ in range(int(rng.integers(1, 7))):
        start = int(rng.integers(0, n))
        count = int(rng.integers(0, n - start + 1))
        spp = int(rng.integers(0, 6))
        if spp == 0: # no-op on the GPU by contract; the reference's running mean would write 0/0 into the range
            dev.render(film, pt.make_job(start, count, 0, offset, 1000 + seed))
            continue
        job = pt.make_job(start, count, spp, offset, 1000 + seed)
        offset += spp
        dev.render(film, job)
        dev.render_host(job, h_img, h_pos, h_nrm, h_cnt)
        osc.render(job, img, pos, nrm, cnt, threads=4)
    g_img, g_pos, g_nrm, g_cnt = film.download()
    same_host = all(np.array_equal(a, b, equal_nan=True) for a, b in ((g_img, h_img), (g_pos, h_pos), (g_nrm, h_nrm), (g_cnt, h_cnt)))
    close = np.isclose(g_img, img, rtol=1e-9, atol=1e-12, equal_nan=True).all(axis=2).mean()
    ok = same_host and np.array_equal(g_cnt, cnt) and close >= 0.99 and np.allclose(g_pos, pos, rtol=1e-9, atol=1e-9, equal_nan=True)
    failures += not ok
    print(seed, "ok" if ok else f"FAILED host-entry-same={same_host} counters={np.array_equal(g_cnt, cnt)} close={close:.4f}", flush=True)
print("failures:", failures)
