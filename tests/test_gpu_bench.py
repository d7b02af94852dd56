"""bench.py on a GPU box: the output contract, and the multi-GPU code path rehearsed with one rank."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--width", "320", "--height", "180", "--steps", "5", "--warmup", "1", "--cpu-threads", "4", "--mesh-lon", "64",
         "--mesh-lat", "33", "--mesh-steps", "2", "--tail-records", "0"]  # tail_records 0: every bounce through k_bounce / k_traverse


def check_line(out, n_gpus):
    line = [l for l in out.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == n_gpus and d["steps"] == 5 and d["warmup"] == 1
    assert d["unit"] == "Mpaths/s" and d["value"] > 0 and d["scaling"] == "weak" and d["dtype"] == "f64"
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and "workload" in d["config"]
    check_roofline(d["roofline"])
    m = d["mesh"]  # the other half of the metric, measured in the same process
    assert m["value"] > 0 and m["unit"] == "Mpaths/s" and "triangle OBJ mesh" in m["config"]["workload"]
    assert m["rays"]["nodes_per_traversed_ray"] > 0 and m["rays"]["traversal_stack_overflows"] == 0
    check_roofline(m["roofline"])
    for record in (d, m):  # what an efficiency loss at N > 1 would be attributed with
        ranks = record["ranks"]
        assert ranks["world_size"] == n_gpus and ranks["ms_per_step_max"] >= ranks["ms_per_step_min"] > 0
        assert ranks["combine_ms_max"] >= 0 and ranks["combine_bytes_per_rank"] >= 0
    return d


def check_roofline(r):
    assert (r["bound"], r["unit"], r["peak"]) in (("hbm", "GB/s", 8000.0), ("valu_f64", "TFLOP/s", 78.6))
    assert r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["kernel"].startswith("ptk::k_") and r["profiled_steps"] >= 1
    assert r["hbm"]["algorithmic_GBps"] > 0 and r["ms_per_launch"] > 0


def test_default_path_prints_the_contract_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = check_line(r.stdout, 1)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 4 and c["value"] > 0 and c["unit"] == "Mpaths/s" and c["sample"]
    assert d["denoise"]["ms_per_frame"] > 0 and d["denoise"]["roofline"]["bound"] == "hbm"


def test_multi_gpu_path_with_one_rank():
    env = dict(os.environ, PTK_BENCH_FORCE_DISTRIBUTED="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no-cpu-baseline"] + SMALL
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = check_line(r.stdout, 1)
    assert d["cpu_baseline"] is None
    # one all-reduce of the three films and the weights: 10 doubles per pixel
    assert d["ranks"]["backend"] == "nccl" and d["ranks"]["combine_ms_max"] > 0
    assert d["ranks"]["combine_bytes_per_rank"] == 10 * 8 * 320 * 180


def test_tile_partition_with_one_rank():
    env = dict(os.environ, PTK_BENCH_FORCE_DISTRIBUTED="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29518", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no-cpu-baseline", "--partition", "tiles"] + SMALL
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = check_line(r.stdout, 1)
    assert "bands" in d["config"]["partition"] and "all-gather" in d["config"]["partition"]
    # a rank contributes its own bands: 23 bands of 8 rows cover the 180 rows, 9 doubles and a counter per pixel
    assert d["ranks"]["backend"] == "nccl" and d["ranks"]["combine_bytes_per_rank"] == (9 * 8 + 8) * 320 * 184
