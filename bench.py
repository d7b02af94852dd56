#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric: Mpaths/s (whole job).  A path is one camera sample (one `raytrace` call of the reference,
app/pathtracer.cpp:39).  A step renders `--spp-per-step` samples of every pixel of the frame and
folds them into the device-resident film; scene, path pools and film are in HBM before the timed
region starts.  At N > 1 (launched by torch.distributed.run, one rank per GPU) every rank renders
its own slice of sample indices of the whole frame -- per-GPU work fixed, weak scaling -- and the
films are summed over RCCL once at the end of the timed region.

Workloads (`--workload`):
  example   the reference's scenes/example Cornell box at 1920x1080 (benchmark configuration 2)
  mesh      Cornell walls + a generated 1,048,576-triangle OBJ mesh at 1920x1080
  cloud     scattering-medium + coloured-glass scene at 1920x1080
  room      living-room stand-in (23 objects, textured OBJ meshes, file textures, sphere light) at 1920x1080
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="example", choices=["example", "mesh", "cloud", "room"])
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp-per-step", type=int, default=8)
    ap.add_argument("--mesh-lon", type=int, default=1024)
    ap.add_argument("--mesh-lat", type=int, default=513)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--denoise", action="store_true", help="also time the shipped denoising pipeline on the device film")
    ap.add_argument("--tail-records", type=int, default=-1, help="override the library's tail_records knob")
    ap.add_argument("--bvh-leaf-size", type=int, default=0, help="override the scene setting bvh_leaf_size (mesh workload)")
    return ap.parse_args()


def build_workload(args, directory):
    from pathtracing_amd import scenes
    settings = dict(WIDTH=args.width, HEIGHT=args.height, samples_per_pixel=args.spp_per_step * args.steps,
                    samples_per_iteration=args.spp_per_step)
    if args.bvh_leaf_size > 0:
        settings["bvh_leaf_size"] = args.bvh_leaf_size
    if args.workload == "example":
        scenes.cornell_example(directory, **settings)
        return "scenes/example Cornell box (12 analytic objects)"
    if args.workload == "mesh":
        n = scenes.mesh_box(directory, n_lon=args.mesh_lon, n_lat=args.mesh_lat, material="gold", smooth=True, **settings)
        return f"Cornell walls + generated {n}-triangle OBJ mesh (MetallicMicrofacet)"
    if args.workload == "room":
        scenes.living_room(directory, mesh_lon=256, mesh_lat=129, texture_size=1024, **settings)
        return "living-room stand-in: 23 objects, 5 OBJ meshes (294,912 triangles), 1024^2 file textures, sphere light"
    scenes.cloud_box(directory, **settings)
    return "Cornell box + homogeneous scattering cloud + coloured glass"


def scenes_default_pipeline():
    from pathtracing_amd import scenes
    return [dict(mode=t["mode"], **t["parameters"]) for t in scenes.DEFAULT_DENOISING["pipeline"]]


def cpu_baseline(host_scene, args, threads):
    """The CPU restatement (oracle/) on this box's host cores, on a bounded sample of the same frame."""
    from oracle import binding as ob
    import pathtracing_amd as pt
    W, H = args.width, args.height
    osc = ob.OracleScene(host_scene)
    image, pos, nrm = np.zeros((H, W, 3)), np.zeros((H, W, 3)), np.zeros((H, W, 3))
    cnt = np.zeros((H, W), dtype=np.uint64)
    # calibrate on a thin slice, then size the sample for roughly 12 s
    rows = max(1, H // 135)
    job = pt.make_job(0, rows * W, 1, 0, args.seed)
    t0 = time.perf_counter()
    osc.render(job, image, pos, nrm, cnt, threads=threads)
    rate = rows * W / max(time.perf_counter() - t0, 1e-9)
    # the thin slice underestimates the rate (thread start-up, one region of the frame): a whole frame at a sample
    # count sized for about a second gives the figure the final sample is sized with
    spp = int(max(1, min(64, rate * 1.0 / (W * H))))
    job = pt.make_job(0, W * H, spp, 0, args.seed)
    t0 = time.perf_counter()
    osc.render(job, image, pos, nrm, cnt, threads=threads)
    rate = max(rate, W * H * spp / max(time.perf_counter() - t0, 1e-9))
    budget = rate * 12.0  # paths
    spp = int(max(1, min(256, budget / (W * H))))
    n_pixels = W * H if budget >= W * H else int(max(W, budget // W * W))
    job = pt.make_job(0, n_pixels, spp, 1, args.seed)
    t0 = time.perf_counter()
    osc.render(job, image, pos, nrm, cnt, threads=threads)
    dt = time.perf_counter() - t0
    paths = n_pixels * spp
    return {"value": paths / dt / 1e6, "unit": "Mpaths/s", "cores": threads, "kind": "port",
            "sample": f"{n_pixels} pixels x {spp} spp of the same {W}x{H} frame ({paths} paths, {dt:.1f} s), "
                      "fp64 CPU restatement of the reference with the same counter-based random streams"}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # PTK_BENCH_FORCE_DISTRIBUTED=1 takes the multi-GPU path (torch tensors, RCCL all-reduce) with any world
    # size, so that it can be rehearsed on a one-GPU box
    distributed = world > 1 or os.environ.get("PTK_BENCH_FORCE_DISTRIBUTED") == "1"
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    import pathtracing_amd as pt

    pt.init(local_rank)
    if args.tail_records >= 0:
        pt.set_option("tail_records", args.tail_records)
    tmp = tempfile.mkdtemp(prefix=f"ptk_bench_r{rank}_")
    workload_name = build_workload(args, tmp)
    t_build = time.perf_counter()
    host_scene = pt.load_scene_directory(tmp)
    W, H = args.width, args.height
    n_pixels = W * H
    spp = args.spp_per_step
    scene = pt.DeviceScene(host_scene, max_paths_in_flight=max(n_pixels * spp, 1 << 22))
    build_s = time.perf_counter() - t_build

    if distributed:
        import torch
        dev = torch.device("cuda", local_rank)
        t_img = torch.zeros(n_pixels * 3, dtype=torch.float64, device=dev)
        t_pos = torch.zeros(n_pixels * 3, dtype=torch.float64, device=dev)
        t_nrm = torch.zeros(n_pixels * 3, dtype=torch.float64, device=dev)
        t_cnt = torch.zeros(n_pixels, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        film = pt.Film(W, H, device_buffers=(t_img.data_ptr(), t_pos.data_ptr(), t_nrm.data_ptr(), t_cnt.data_ptr()))
    else:
        film = pt.Film(W, H)

    def barrier():
        pt.sync()
        if distributed:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    # every rank owns a disjoint slice of sample indices (weak scaling: the slice length is fixed)
    from pathtracing_amd.distributed import combine_films, sample_slice
    total_spp = spp * (args.warmup + args.steps)
    offset, _ = sample_slice(rank, world, total_spp * world)

    # Kernels are timed with HIP events on the library's stream on every PROFILE_EVERY-th step of the
    # timed region (event records drain the queue, so timing every step would slow the run).
    PROFILE_EVERY = 4
    CLASS_NAMES = ["Diffuse", "Reflective", "Transparent", "Glossy", "MetallicMicrofacet", "ReflectiveMicrofacet",
                   "TransparentMicrofacet", "MediumScatter"]
    profiled = {"steps": 0}
    fam = {"k_bounce": [0.0, 0, 0], "k_bin": [0.0, 0, 0], "k_traverse": [0.0, 0, 0]}  # ms, bytes, launches
    other_ms = [0.0]

    def step(i, timed=False):
        job = pt.make_job(0, n_pixels, spp, offset + i * spp, args.seed)
        if timed and (i - args.warmup) % PROFILE_EVERY == 0:
            s0 = scene.stats()
            pt.set_profiling(True)
            scene.render(film, job)
            pt.set_profiling(False)
            s1 = scene.stats()
            profiled["steps"] += 1
            for k, f in (("k_bounce", "bounce"), ("k_bin", "bin"), ("k_traverse", "traverse")):
                fam[k][0] += getattr(s1, f + "_ms") - getattr(s0, f + "_ms")
                fam[k][1] += getattr(s1, f + "_bytes") - getattr(s0, f + "_bytes")
                fam[k][2] += getattr(s1, f + "_launches") - getattr(s0, f + "_launches")
            other_ms[0] += s1.other_ms - s0.other_ms
        else:
            scene.render(film, job)

    for i in range(args.warmup):
        step(i)
    if distributed and args.warmup > 0:
        # the warm-up also runs the combine once: RCCL sets up its channels and torch's allocator gets its
        # blocks at the first large collective, which would otherwise fall inside the timed region
        pt.sync()
        combine_films(t_img, t_pos, t_nrm, t_cnt)
        torch.cuda.synchronize()
    barrier()
    film.clear()
    scene.reset_stats()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i, timed=True)
    pt.sync()
    if distributed:
        import torch
        # combine the per-GPU sample slices: weighted mean over RCCL, once per flush
        combine_films(t_img, t_pos, t_nrm, t_cnt)
        torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        import torch
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=torch.device("cuda", local_rank))
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    st = scene.stats()
    paths_per_rank = n_pixels * spp * args.steps
    total_paths = paths_per_rank * world
    value = total_paths / elapsed / 1e6

    # The dominant kernel, timed with HIP events on the library's stream over the profiled steps of this timed
    # region and named as rocprofv3 shows it: k_bounce (every path vertex: shading, shadow ray, extension ray; in
    # scenes with meshes its time includes the resume pass) or, in scenes with meshes, k_traverse.
    if fam["k_traverse"][0] > fam["k_bounce"][0]:
        kernel = "ptk::k_traverse"
        d_ms, d_bytes, d_launches = fam["k_traverse"]
    else:
        kernel = scene.bounce_kernel()
        d_ms, d_bytes, d_launches = fam["k_bounce"]
    achieved = d_bytes / (d_ms * 1e-3) / 1e9 if d_ms > 0 else 0.0
    traffic = None
    traffic_file = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(traffic_file):
        try:
            with open(traffic_file) as f:
                traffic = json.load(f).get(args.workload, {}).get(kernel)
        except Exception:
            traffic = None
    # SQ counters of the same kernel from the last rocprofv3 --pmc passes kept under profiles/ (tools/counters_digest.py):
    # the share of the SIMDs' issue slots spent on vector instructions and the occupancy the kernel is compiled for
    valu = None
    counters_file = os.path.join(ROOT, "profiles", "counters_latest.json")
    if os.path.exists(counters_file):
        try:
            with open(counters_file) as f:
                c = json.load(f).get(args.workload, {}).get(kernel)
            if c and "valu_share_of_wave_time" in c:
                waves_per_simd = 5 if kernel == "ptk::k_traverse" else 2
                valu = {"waves_per_simd": waves_per_simd, "occupancy_frac_of_8": waves_per_simd / 8.0,
                        "simd_issue_busy": c["valu_share_of_wave_time"] * waves_per_simd,
                        "active_lanes_per_instruction": c.get("active_lanes_per_valu"),
                        "valu_instructions_per_wave": c.get("valu_per_wave"), "source": "profiles/counters_latest.json"}
        except Exception:
            valu = None
    vertices_by_class = {CLASS_NAMES[c]: int(st.class_vertices[c]) for c in range(8) if st.class_vertices[c]}
    roofline = {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "valu": valu,
                "bytes_per_launch": d_bytes / max(d_launches, 1), "ms_per_launch": d_ms / max(d_launches, 1),
                "launches": int(d_launches), "profiled_steps": profiled["steps"],
                "device_ms": {"k_bounce": fam["k_bounce"][0], "k_bin": fam["k_bin"][0], "k_traverse": fam["k_traverse"][0],
                              "other": other_ms[0]},
                "vertices_by_class": vertices_by_class}

    out = {
        "metric": "Mpaths/s (whole node) at 1080p, scenes/example + 1M-tri mesh, 1/2/4/8 GPUs",
        "value": value, "unit": "Mpaths/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{workload_name}, {W}x{H}, {spp} spp per step ({spp * args.steps} spp in the timed region)",
                   "paths_per_step_per_gpu": n_pixels * spp, "partition": "sample-index slices per GPU, film all-reduce at the end",
                   "scene_build_s": build_s},
        "rays": {"Grays_per_s": (st.extension_rays + st.shadow_rays) * world / elapsed / 1e9,
                 "extension_rays_per_path": st.extension_rays / max(st.paths, 1),
                 "shadow_rays_per_path": st.shadow_rays / max(st.paths, 1),
                 "vertices_per_path": st.vertices / max(st.paths, 1),
                 "algorithmic_bytes_per_path": st.algorithmic_bytes / max(st.paths, 1),
                 "traversed_rays_per_path": st.traversed_rays / max(st.paths, 1),
                 "nodes_per_traversed_ray": st.nodes_visited / max(st.traversed_rays, 1),
                 "triangles_per_traversed_ray": st.triangles_tested / max(st.traversed_rays, 1)},
        "roofline": roofline,
    }
    if args.denoise and not distributed:
        # the reference's shipped pipeline (denoising.json: median 3x3 threshold 0.5, then 4 a-trous iterations) on the
        # rendered film, device to device; bytes: 48 B/pixel for the median, 96 B/pixel per a-trous iteration
        tasks = pt.make_denoise_tasks(scenes_default_pipeline())
        scratch = pt.Film(W, H)
        pt.denoise_film(tasks, film, scratch)
        pt.sync()
        reps = 10
        t1 = time.perf_counter()
        for _ in range(reps):
            pt.denoise_film(tasks, film, scratch)
        pt.sync()
        dt = (time.perf_counter() - t1) / reps
        out["denoise"] = {"pipeline": "median 3x3 + a-trous x4 (the reference's denoising.json)", "ms_per_frame": dt * 1e3,
                          "Mpixels_per_s": n_pixels / dt / 1e6, "algorithmic_GBps": n_pixels * (48 + 4 * 96) / dt / 1e9}
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        threads = args.cpu_threads or min(16, os.cpu_count() or 1)
        out["cpu_baseline"] = cpu_baseline(host_scene, args, threads)
        out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
