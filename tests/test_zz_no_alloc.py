"""ptk_denoise_device_static, the entry point that allocates nothing (its ping-pong buffer is a static array in
device memory), against ptk_denoise_device with a caller-provided scratch buffer: same kernels, so the outputs must
be bit-identical.  Two calls in a row, the second on a smaller image (the static buffer then still holds the first
call's data), then the size limit."""
import ctypes as C

import numpy as np
import pytest

import pathtracing_amd as pt
from pathtracing_amd import _ffi

pytestmark = pytest.mark.gpu

PIPELINE = [{"mode": "median", "kernel_size": 3, "threshold": 0.5},
            {"mode": "atrous", "iterations": 3, "sigma_rt": 2, "sigma_x": 2, "sigma_n": 2},
            {"mode": "median", "kernel_size": 5, "threshold": 0.05}]


def _frame(width, height, seed):
    rng = np.random.default_rng(seed)
    image = rng.random((height, width, 3)) * 2.0
    image[rng.random((height, width)) < 0.02] = 40.0  # fireflies for the median to remove
    position = rng.normal(size=(height, width, 3))
    normal = rng.normal(size=(height, width, 3))
    normal /= np.linalg.norm(normal, axis=2, keepdims=True)
    return image, position, normal


def _both_ways(width, height, seed):
    lib = pt.device_lib()
    tasks = pt.make_denoise_tasks(PIPELINE)
    image, position, normal = _frame(width, height, seed)
    outputs = []
    for static in (False, True):
        film = pt.Film(width, height)
        film.upload(image, position, normal)
        img, pos, nrm, _ = [C.c_void_p(p) for p in film.device_pointers()]
        if static:
            rc = lib.ptk_denoise_device_static(tasks, len(tasks), width, height, img, pos, nrm)
        else:
            scratch = pt.Film(width, height)
            rc = lib.ptk_denoise_device(tasks, len(tasks), width, height, img, pos, nrm, C.c_void_p(scratch.device_pointers()[0]))
        assert rc == 0, lib.ptk_last_error()
        pt.sync()
        outputs.append(film.download()[0])
    return image, outputs


def test_static_scratch_entry_point_matches_the_allocating_one_twice_in_a_row():
    pt.init(0)
    for width, height, seed in ((160, 96, 1), (72, 40, 2)):  # the second call is smaller than the first
        image, (want, got) = _both_ways(width, height, seed)
        assert not np.array_equal(want, image)  # the pipeline did something
        assert np.array_equal(want, got), (width, height)


def _free_device_memory():
    """Free device memory as the HIP runtime this process already has loaded reports it."""
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def test_static_scratch_entry_point_allocates_no_device_memory():
    """The property the entry point exists for: free device memory is the same before and after two calls (one on a
    larger, one on a smaller image), while the allocating entry point's caller has to bring a scratch buffer."""
    pt.init(0)
    lib = pt.device_lib()
    tasks = pt.make_denoise_tasks(PIPELINE)
    frames = []
    for width, height, seed in ((256, 144, 3), (96, 64, 4)):
        film = pt.Film(width, height)
        film.upload(*_frame(width, height, seed))
        frames.append((width, height, film))
    # once before measuring: whatever the runtime sets up lazily at a first launch of these kernels is in place
    for width, height, film in frames:
        img, pos, nrm, _ = [C.c_void_p(p) for p in film.device_pointers()]
        assert lib.ptk_denoise_device_static(tasks, len(tasks), width, height, img, pos, nrm) == 0
    pt.sync()
    before = _free_device_memory()
    for width, height, film in frames:
        img, pos, nrm, _ = [C.c_void_p(p) for p in film.device_pointers()]
        assert lib.ptk_denoise_device_static(tasks, len(tasks), width, height, img, pos, nrm) == 0
        assert _free_device_memory() == before  # (while the kernels may still be running)
    pt.sync()
    assert _free_device_memory() == before


def test_static_scratch_entry_point_refuses_images_beyond_its_buffer():
    pt.init(0)
    lib = pt.device_lib()
    tasks = pt.make_denoise_tasks(PIPELINE)
    film = pt.Film(8, 8)  # the pointers are never used: the size check comes first
    img, pos, nrm, _ = [C.c_void_p(p) for p in film.device_pointers()]
    rc = lib.ptk_denoise_device_static(tasks, len(tasks), 2048, 1025, img, pos, nrm)
    assert rc == -3  # PTK_ERR_MEMORY
    assert b"2097152" in lib.ptk_last_error()
    assert lib.ptk_denoise_device_static(tasks, len(tasks), 0, 4, img, pos, nrm) == -1
