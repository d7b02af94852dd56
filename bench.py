#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric: Mpaths/s (whole job).  A path is one camera sample (one `raytrace` call of the reference,
app/pathtracer.cpp:39).  A step renders `--spp-per-step` samples of every pixel of the frame and
folds them into the device-resident film; scene, path pools and film are in HBM before the timed
region starts.  At N > 1 (launched by torch.distributed.run, one rank per GPU) the work of a step is
partitioned with per-GPU work fixed (weak scaling) in one of two ways (`--partition`):

  samples  every rank renders its own slice of sample indices of the whole frame; the films are combined
           with one weighted-mean all-reduce (RCCL) at the end of the timed region
  tiles    the frame is N times as tall; every rank owns the interleaved bands of rows `distributed.row_bands`
           gives it and renders only those pixels; one all-gather of every rank's own bands (1/N of the frame
           each) assembles the frame on every rank

The line's "ranks" record holds what an efficiency loss at N > 1 would be attributed with: the slowest and the
fastest rank's time per step, the time of the combine, the bytes a rank contributes to it, and the world size and
backend as torch.distributed reports them.

`value` is the headline workload (`--workload`, default the reference's scenes/example at 1920x1080).  The default
run also measures, in the same process, the other half of the metric -- a generated 1,048,576-triangle mesh in the
Cornell box at 1080p, reported under "mesh" -- and the reference's shipped denoising pipeline on a frame of the
scattering-medium scene, reported under "denoise" (`--only-headline` skips both).

Workloads:
  example   the reference's scenes/example Cornell box (benchmark configuration 2)
  mesh      Cornell walls + a generated 1,048,576-triangle OBJ mesh (MetallicMicrofacet)
  cloud     scattering-medium + coloured-glass scene
  room      living-room stand-in (23 objects, textured OBJ meshes, file textures, sphere light)
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0    # MI355X HBM3E peak
F64_PEAK_TFLOPS = 78.6   # MI355X fp64 vector peak (AMD's specification; half the 157.3 TFLOP/s fp32 vector rate)
CLASS_NAMES = ["Diffuse", "Reflective", "Transparent", "Glossy", "MetallicMicrofacet", "ReflectiveMicrofacet",
               "TransparentMicrofacet", "MediumScatter"]
PROFILE_EVERY = 16  # HIP events bracket the kernel families on every sixteenth timed step (the first one, in a default run): event
                    # records drain the queue, and a profiled step traces one batch at a time (a launch's time is its own
                    # only then), which makes it 6 % slower on the example and 16 % slower on the mesh (three batches in flight)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="example", choices=["example", "mesh", "cloud", "room", "example_deferred"])
    ap.add_argument("--partition", default="samples", choices=["samples", "tiles"])
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp-per-step", type=int, default=64)
    ap.add_argument("--mesh-lon", type=int, default=1024)
    ap.add_argument("--mesh-lat", type=int, default=513)
    ap.add_argument("--mesh-steps", type=int, default=16, help="timed steps of the mesh sub-record")
    ap.add_argument("--only-headline", action="store_true", help="skip the mesh and denoise sub-records")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE", help="library tuning knob (ptk_set_option)")
    ap.add_argument("--tail-records", type=int, default=-1, help="override the library's tail_records knob")
    ap.add_argument("--bvh-leaf-size", type=int, default=0, help="override the scene setting bvh_leaf_size (mesh workload)")
    return ap.parse_args()


def sources_hash():
    """Hash of the device library's sources: profiles kept under profiles/ record it, so a counter file measured on
    other code than the one running is recognised (there is no git on the GPU box)."""
    h = hashlib.sha256()
    files = []
    for pattern in ("*.hip", "*.h", "*/*.hip", "*/*.h", "*/*.cpp", "*/*/*.hip", "*/*/*.h"):
        files += glob.glob(os.path.join(ROOT, "pathtracing_amd", "csrc", pattern))
    for path in sorted(set(files)):
        if os.sep + "build" + os.sep in path:
            continue
        h.update(os.path.relpath(path, ROOT).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def build_workload(args, workload, directory, width, height, spp, steps):
    from pathtracing_amd import scenes
    settings = dict(WIDTH=width, HEIGHT=height, samples_per_pixel=spp * steps, samples_per_iteration=spp)
    if args.bvh_leaf_size > 0:
        settings["bvh_leaf_size"] = args.bvh_leaf_size
    if workload == "example":
        scenes.cornell_example(directory, **settings)
        return "scenes/example Cornell box (12 analytic objects)"
    if workload == "example_deferred":
        scenes.cornell_example_all_deferred(directory, **settings)
        return "scenes/example with every vertex sent through the deferred route (a two-triangle mesh whose bounds hold the room)"
    if workload == "mesh":
        n = scenes.mesh_box(directory, n_lon=args.mesh_lon, n_lat=args.mesh_lat, material="gold", smooth=True, **settings)
        return f"Cornell walls + generated {n}-triangle OBJ mesh (MetallicMicrofacet)"
    if workload == "room":
        scenes.living_room(directory, mesh_lon=256, mesh_lat=129, texture_size=1024, **settings)
        return "living-room stand-in: 23 objects, 5 OBJ meshes (294,912 triangles), 1024^2 file textures, sphere light"
    scenes.cloud_box(directory, **settings)
    return "Cornell box + homogeneous scattering cloud + coloured glass"


def scenes_default_pipeline():
    from pathtracing_amd import scenes
    return [dict(mode=t["mode"], **t["parameters"]) for t in scenes.DEFAULT_DENOISING["pipeline"]]


def cpu_baseline(host_scene, width, height, seed, threads):
    """The CPU restatement (oracle/) on this box's host cores, on a bounded sample of the same frame."""
    from oracle import binding as ob
    import pathtracing_amd as pt
    W, H = width, height
    osc = ob.OracleScene(host_scene)
    image, pos, nrm = np.zeros((H, W, 3)), np.zeros((H, W, 3)), np.zeros((H, W, 3))
    cnt = np.zeros((H, W), dtype=np.uint64)
    # calibrate on a thin slice, then size the sample for roughly 12 s
    rows = max(1, H // 135)
    job = pt.make_job(0, rows * W, 1, 0, seed)
    t0 = time.perf_counter()
    osc.render(job, image, pos, nrm, cnt, threads=threads)
    rate = rows * W / max(time.perf_counter() - t0, 1e-9)
    # the thin slice underestimates the rate (thread start-up, one region of the frame): a whole frame at a sample
    # count sized for about a second gives the figure the final sample is sized with
    spp = int(max(1, min(64, rate * 1.0 / (W * H))))
    job = pt.make_job(0, W * H, spp, 0, seed)
    t0 = time.perf_counter()
    osc.render(job, image, pos, nrm, cnt, threads=threads)
    rate = max(rate, W * H * spp / max(time.perf_counter() - t0, 1e-9))
    budget = rate * 12.0  # paths
    spp = int(max(1, min(256, budget / (W * H))))
    n_pixels = W * H if budget >= W * H else int(max(W, budget // W * W))
    job = pt.make_job(0, n_pixels, spp, 1, seed)
    t0 = time.perf_counter()
    osc.render(job, image, pos, nrm, cnt, threads=threads)
    dt = time.perf_counter() - t0
    paths = n_pixels * spp
    return {"value": paths / dt / 1e6, "unit": "Mpaths/s", "cores": threads, "kind": "port",
            "sample": f"{n_pixels} pixels x {spp} spp of the same {W}x{H} frame ({paths} paths, {dt:.1f} s), "
                      "fp64 CPU restatement of the reference with the same counter-based random streams"}


def profile_entry(name, workload, kernel, code):
    """Per-kernel figures of the last rocprofv3 counter passes kept under profiles/ (tools/profile_round.sh,
    tools/pmc_pass.sh and their digests), with whether they were measured on the code that is running."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, None
    try:
        with open(path) as f:
            data = json.load(f)
        entry = data.get(workload, {}).get(kernel)
        measured_on = data.get("_sources", {}).get(workload)
        return entry, {"file": "profiles/" + name, "measured_on_sources": measured_on, "current_sources": code,
                       "stale": measured_on != code}
    except Exception:
        return None, None


class Distributed:
    """torch.distributed (RCCL) state of this rank; inert in a single-process run."""

    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        # PTK_BENCH_FORCE_DISTRIBUTED=1 takes the multi-GPU path (torch tensors, RCCL collectives) with any world
        # size, so that it can be rehearsed on a one-GPU box
        self.on = self.world > 1 or os.environ.get("PTK_BENCH_FORCE_DISTRIBUTED") == "1"
        if self.on:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29512")
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            torch.cuda.set_device(self.local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", self.local_rank), rank=self.rank,
                                    world_size=self.world)

    def barrier(self, pt):
        pt.sync()
        if self.on:
            self.torch.cuda.synchronize()
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        return self.over_ranks(seconds, "MAX")

    def over_ranks(self, value, op):
        """MIN, MAX or SUM of a per-rank number."""
        if not self.on:
            return value
        t = self.torch.tensor([value], dtype=self.torch.float64, device=self.torch.device("cuda", self.local_rank))
        self.dist.all_reduce(t, op=getattr(self.dist.ReduceOp, op))
        return float(t.item())

    def describe(self):
        """What the collective library itself reports about the job."""
        if not self.on:
            return {"world_size": 1, "backend": None}
        return {"world_size": self.dist.get_world_size(), "backend": self.dist.get_backend(), "rank_env": self.world}

    def close(self):
        if self.on:
            self.dist.destroy_process_group()


def run_workload(pt, args, ds, workload, width, height, spp, steps, warmup, code):
    """W untimed steps, then exactly K timed ones between barriers; returns the record of the workload."""
    from pathtracing_amd.distributed import FilmCombiner, band_job, sample_slice
    rank, world = ds.rank, ds.world
    tiles = args.partition == "tiles"
    frame_height = height * world if tiles else height  # tiles: per-GPU pixel count fixed, the frame grows
    tmp = tempfile.mkdtemp(prefix=f"ptk_bench_{workload}_r{rank}_")
    workload_name = build_workload(args, workload, tmp, width, frame_height, spp, steps)
    t_build = time.perf_counter()
    host_scene = pt.load_scene_directory(tmp)
    W, H = width, frame_height
    n_pixels = W * H
    band_start, own_pixels, band_pixels, band_stride = band_job(rank, world, W, H, band_rows=8) if tiles else (0, n_pixels, 0, 0)
    # a step of more than 2^24 paths is traced as batches of whole pixels, several at a time (option batch_lanes), each of
    # at most 2^24 paths
    paths_in_flight = max(min(width * height * spp, max(int(dict(o.split("=", 1) for o in args.option).get("batch_lanes", 0)), 3) << 24), 1 << 22)
    scene = pt.DeviceScene(host_scene, max_paths_in_flight=paths_in_flight)
    build_s = time.perf_counter() - t_build

    combiner = None
    if ds.on:
        # the films live in one torch tensor (FilmCombiner) that the library renders into; every buffer the combine
        # needs is allocated here, before the timed region
        torch = ds.torch
        combiner = FilmCombiner(n_pixels, torch.device("cuda", ds.local_rank), world, mode=args.partition,
                                band_pixels=band_pixels if tiles else 0)
        torch.cuda.synchronize()
        film = pt.Film(W, H, device_buffers=(combiner.image.data_ptr(), combiner.position.data_ptr(), combiner.normal.data_ptr(),
                                             combiner.counters.data_ptr()))
    else:
        film = pt.Film(W, H)

    # samples: every rank owns a disjoint slice of sample indices (weak scaling: the slice length is fixed)
    total_spp = spp * (warmup + steps)
    offset = 0 if tiles else sample_slice(rank, world, total_spp * world)[0]

    profiled = {"steps": 0}
    batch_lanes = int(dict(o.split("=", 1) for o in args.option).get("batch_lanes", 0))
    fam = {"k_bounce": [0.0, 0, 0], "k_bin": [0.0, 0, 0], "k_traverse": [0.0, 0, 0]}  # ms, bytes, launches
    other_ms = [0.0]

    def render_step(i):
        if tiles:  # the interleaved 8-row bands of this rank as one banded job
            scene.render_bands(film, pt.make_job(band_start, own_pixels, spp, i * spp, args.seed), band_pixels, band_stride)
        else:
            scene.render(film, pt.make_job(0, n_pixels, spp, offset + i * spp, args.seed))

    def step(i, timed=False):
        if timed and (i - warmup) % PROFILE_EVERY == 0:
            s0 = scene.stats()
            # one batch at a time in a profiled step: with two in flight (scenes with meshes) the kernels of the two
            # share the device, and an event pair around a launch times both
            pt.set_option("batch_lanes", 1)
            pt.set_profiling(True)
            render_step(i)
            pt.set_profiling(False)
            pt.set_option("batch_lanes", batch_lanes)
            s1 = scene.stats()
            profiled["steps"] += 1
            for k, f in (("k_bounce", "bounce"), ("k_bin", "bin"), ("k_traverse", "traverse")):
                fam[k][0] += getattr(s1, f + "_ms") - getattr(s0, f + "_ms")
                fam[k][1] += getattr(s1, f + "_bytes") - getattr(s0, f + "_bytes")
                fam[k][2] += getattr(s1, f + "_launches") - getattr(s0, f + "_launches")
            other_ms[0] += s1.other_ms - s0.other_ms
        else:
            render_step(i)

    for i in range(warmup):
        step(i)
    if ds.on and warmup > 0:
        # the warm-up also runs the combine once: RCCL sets up its channels and torch's allocator gets its
        # blocks at the first large collective, which would otherwise fall inside the timed region
        pt.sync()
        combiner.combine(rank)
        ds.torch.cuda.synchronize()
    ds.barrier(pt)
    film.clear()
    scene.reset_stats()
    ds.barrier(pt)
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i, timed=True)
    pt.sync()
    render_s = time.perf_counter() - t0  # this rank's K steps
    combine_s = 0.0
    if ds.on:
        # combine the per-GPU films over RCCL, once per flush: an all-reduce (sample slices) or an all-gather of every
        # rank's own bands (tiles)
        t1 = time.perf_counter()
        combiner.combine(rank)
        ds.torch.cuda.synchronize()
        combine_s = time.perf_counter() - t1
    ds.barrier(pt)
    elapsed = ds.max_over_ranks(time.perf_counter() - t0)
    ranks = {"ms_per_step_min": ds.over_ranks(render_s, "MIN") / steps * 1e3, "ms_per_step_max": ds.over_ranks(render_s, "MAX") / steps * 1e3,
             "combine_ms_max": ds.over_ranks(combine_s, "MAX") * 1e3,
             "combine_bytes_per_rank": combiner.wire_bytes_per_rank() if combiner else 0}
    ranks.update(ds.describe())

    st = scene.stats()
    total_paths = own_pixels * spp * steps * world
    value = total_paths / elapsed / 1e6

    # The dominant kernel, timed with HIP events on the library's stream over the profiled steps of this timed
    # region and named as rocprofv3 shows it: k_bounce (every path vertex: shading, shadow ray, extension ray) or, in
    # scenes with meshes, k_traverse when it takes longer.
    traverse = fam["k_traverse"][0] > fam["k_bounce"][0]
    kernel = "ptk::k_traverse" if traverse else scene.bounce_kernel()
    d_ms, d_bytes, d_launches = fam["k_traverse"] if traverse else fam["k_bounce"]
    ms_per_launch = d_ms / max(d_launches, 1)
    algorithmic_gbs = d_bytes / (d_ms * 1e-3) / 1e9 if d_ms > 0 else 0.0
    traffic, traffic_source = profile_entry("traffic_latest.json", workload, kernel, code)
    counters, counters_source = profile_entry("counters_latest.json", workload, kernel, code)
    hbm = {"algorithmic_GBps": algorithmic_gbs, "algorithmic_frac_of_peak": algorithmic_gbs / HBM_PEAK_GBS,
           "algorithmic_bytes_per_launch": d_bytes / max(d_launches, 1), "peak_GBps": HBM_PEAK_GBS}
    if traffic and ms_per_launch > 0:
        # measured HBM bytes per launch (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE, separate passes) over the live launch time
        hbm["measured_bytes_per_launch"] = traffic
        hbm["measured_GBps"] = traffic / (ms_per_launch * 1e-3) / 1e9
        hbm["measured_frac_of_peak"] = hbm["measured_GBps"] / HBM_PEAK_GBS
        hbm["measured_over_algorithmic"] = traffic / max(d_bytes / max(d_launches, 1), 1)
        hbm["source"] = traffic_source
    valu = None
    flops_per_launch = None
    if counters:
        waves = counters.get("waves", 0)
        lanes = counters.get("active_lanes_per_valu", 64.0)
        per_wave = (2 * counters.get("fma_f64_per_wave", 0) + counters.get("mul_f64_per_wave", 0) +
                    counters.get("add_f64_per_wave", 0) + counters.get("trans_f64_per_wave", 0))
        launches = counters.get("launches")
        if launches:
            flops_per_launch = per_wave * lanes * waves / launches
        valu = {"occupancy_waves_per_simd": counters.get("waves_per_simd"),
                "simd_issue_busy": counters.get("simd_valu_busy"),
                "valu_share_of_wave_time": counters.get("valu_share_of_wave_time"),
                "wait_share_of_wave_time": counters.get("wait_share_of_wave_time"),
                "active_lanes_per_instruction": lanes,
                "valu_instructions_per_wave": counters.get("valu_per_wave"),
                "f64_flops_per_launch": flops_per_launch, "source": counters_source}
    f64_tflops = flops_per_launch / (ms_per_launch * 1e-3) / 1e12 if flops_per_launch and ms_per_launch > 0 else 0.0
    if valu is not None:
        valu["f64_TFLOPps"] = f64_tflops
        valu["f64_frac_of_peak"] = f64_tflops / F64_PEAK_TFLOPS
        valu["f64_peak_TFLOPps"] = F64_PEAK_TFLOPS
    if not traverse and f64_tflops / F64_PEAK_TFLOPS > algorithmic_gbs / HBM_PEAK_GBS:
        # k_bounce is near neither roof; the nearer one is reported.  Here: the fp64 vector rate.
        roofline = {"bound": "valu_f64", "kernel": kernel, "achieved": f64_tflops, "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": f64_tflops / F64_PEAK_TFLOPS, "traffic": traffic}
    elif not traverse:
        # ... here: HBM, by the algorithmic bytes per launch over the live launch time (the measured bytes and the fp64
        # figures are beside it, under hbm and valu)
        roofline = {"bound": "hbm", "kernel": kernel, "achieved": algorithmic_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": algorithmic_gbs / HBM_PEAK_GBS, "traffic": traffic, "basis": "algorithmic bytes per launch"}
    elif traverse and "measured_GBps" in hbm:
        # k_traverse against HBM, by what HBM delivered: the measured bytes per launch (rocprofv3 FETCH_SIZE x 2 +
        # WRITE_SIZE) over the live launch time.  The algorithmic byte count is beside it (hbm.algorithmic_*): it
        # includes node and triangle fetches that L2 and the Infinity Cache serve, so it flatters.
        roofline = {"bound": "hbm", "kernel": kernel, "achieved": hbm["measured_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": hbm["measured_GBps"] / HBM_PEAK_GBS, "traffic": traffic, "basis": "measured HBM bytes per launch"}
    else:
        # no traffic measurement at hand (or k_bounce without a counter file): the algorithmic byte count against HBM
        roofline = {"bound": "hbm", "kernel": kernel, "achieved": algorithmic_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": algorithmic_gbs / HBM_PEAK_GBS, "traffic": traffic, "basis": "algorithmic bytes per launch"}
    roofline.update({"hbm": hbm, "valu": valu, "ms_per_launch": ms_per_launch, "launches": int(d_launches),
                     "profiled_steps": profiled["steps"],
                     "device_ms": {"k_bounce": fam["k_bounce"][0], "k_bin": fam["k_bin"][0],
                                   "k_traverse": fam["k_traverse"][0], "other": other_ms[0]},
                     "vertices_by_class": {CLASS_NAMES[c]: int(st.class_vertices[c]) for c in range(8) if st.class_vertices[c]}})
    partition = {"samples": "sample-index slices per GPU, one all-reduce of the films at the end",
                 "tiles": "interleaved 8-row bands per GPU (the frame is n_gpus times as tall), one all-gather of every GPU's own bands at the end"}[args.partition]
    record = {
        "value": value, "unit": "Mpaths/s", "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
        "config": {"workload": f"{workload_name}, {W}x{H}, {spp} spp per step ({spp * steps} spp in the timed region)",
                   "paths_per_step_per_gpu": own_pixels * spp,
                   "paths_in_flight_limit_per_gpu": paths_in_flight,
                   "batching": "a step is traced as batches of whole pixels, at most 2^24 paths each, two at a time (three in scenes with meshes)",
                   "partition": partition, "scene_build_s": build_s},
        "rays": {"Grays_per_s": (st.extension_rays + st.shadow_rays) * world / elapsed / 1e9,
                 "extension_rays_per_path": st.extension_rays / max(st.paths, 1),
                 "shadow_rays_per_path": st.shadow_rays / max(st.paths, 1),
                 "vertices_per_path": st.vertices / max(st.paths, 1),
                 "algorithmic_bytes_per_path": st.algorithmic_bytes / max(st.paths, 1),
                 "traversed_rays_per_path": st.traversed_rays / max(st.paths, 1),
                 "nodes_per_traversed_ray": st.nodes_visited / max(st.traversed_rays, 1),
                 "triangles_per_traversed_ray": st.triangles_tested / max(st.traversed_rays, 1),
                 "traversal_stack_overflows": int(st.stack_overflows)},
        "roofline": roofline, "ranks": ranks,
    }
    return record, host_scene, scene, film


def denoise_record(pt, film, width, height):
    """The reference's shipped pipeline (denoising.json: median 3x3 threshold 0.5, then 4 a-trous iterations) on the
    rendered film, device to device.  Algorithmic bytes: 48 B/pixel for the median (image in, image out) and 96 B/pixel
    per a-trous iteration (image, position, normal in; image out), fp64."""
    tasks = pt.make_denoise_tasks(scenes_default_pipeline())
    scratch = pt.Film(width, height)
    for _ in range(3):
        pt.denoise_film(tasks, film, scratch)
    pt.sync()
    reps = 50
    t1 = time.perf_counter()
    for _ in range(reps):
        pt.denoise_film(tasks, film, scratch)
    pt.sync()
    dt = (time.perf_counter() - t1) / reps
    n_pixels = width * height
    gbs = n_pixels * (48 + 4 * 96) / dt / 1e9
    return {"pipeline": "median 3x3 + a-trous x4 (the reference's denoising.json)", "ms_per_frame": dt * 1e3,
            "Mpixels_per_s": n_pixels / dt / 1e6,
            "roofline": {"bound": "hbm", "kernel": "ptk::k_atrous", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gbs / HBM_PEAK_GBS,
                         "note": "whole pipeline (5 launches) over its algorithmic bytes; per-kernel times in profiles/"}}


def main():
    args = parse_args()
    ds = Distributed()
    import pathtracing_amd as pt

    pt.init(ds.local_rank)
    if args.tail_records >= 0:
        pt.set_option("tail_records", args.tail_records)
    for item in args.option:
        name, _, value = item.partition("=")
        pt.set_option(name, int(value))
    code = sources_hash()
    record, host_scene, scene, film = run_workload(pt, args, ds, args.workload, args.width, args.height, args.spp_per_step,
                                                   args.steps, args.warmup, code)
    out = {
        "metric": "Mpaths/s (whole node) at 1080p, scenes/example + 1M-tri mesh, 1/2/4/8 GPUs",
        "value": record["value"], "unit": "Mpaths/s", "n_gpus": ds.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": record["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic", "config": record["config"], "rays": record["rays"],
        "roofline": record["roofline"], "ranks": record["ranks"], "device_library_sources": code,
    }
    if ds.rank == 0 and not args.no_cpu_baseline and ds.world == 1:
        threads = args.cpu_threads or min(16, os.cpu_count() or 1)
        out["cpu_baseline"] = cpu_baseline(host_scene, args.width, args.height, args.seed, threads)
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    elif ds.rank == 0:
        out["cpu_baseline"] = None
    del scene, film, host_scene
    if not args.only_headline and args.workload == "example":
        # the other half of the metric: the 1,048,576-triangle mesh scene, same frame size, same partition
        mesh, _, mesh_scene, mesh_film = run_workload(pt, args, ds, "mesh", args.width, args.height, args.spp_per_step,
                                                      args.mesh_steps, 2, code)
        out["mesh"] = mesh
        del mesh_scene, mesh_film
        if not ds.on:
            # benchmark configuration 5: a frame of the scattering-medium scene, then the denoising pipeline on the GPU
            cloud, _, cloud_scene, cloud_film = run_workload(pt, args, ds, "cloud", args.width, args.height, args.spp_per_step,
                                                             4, 1, code)
            out["denoise"] = denoise_record(pt, cloud_film, args.width, args.height)
            out["denoise"]["frame"] = {"workload": cloud["config"]["workload"], "Mpaths_per_s": cloud["value"]}
            del cloud_scene, cloud_film
    if ds.rank == 0:
        print(json.dumps(out))
    ds.close()


if __name__ == "__main__":
    main()
