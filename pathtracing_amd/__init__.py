"""Python front end of the MI355X path-tracing hot path.

Thin ctypes layer over the two C-ABI libraries built from csrc/:

* ``libptkhost.so``  scene-directory loader (settings.json / scene.json / denoising.json, OBJ, .map)
* ``libptk.so``      the HIP kernels behind include/ptk.h

The reference program has no Python API; its user contract is ``pathtracer <scene_directory>`` /
``denoiser <scene_directory>`` (app/pathtracer.cpp:300-380, app/denoiser.cpp:10-54).  The functions
here follow that contract's units: a scene directory, a job of contiguous pixels x samples, three
float64 [H, W, 3] accumulation buffers.  There is no CPU rendering path in this package: every
render/trace/denoise call goes to the GPU library and raises if it is not available.
"""
import ctypes as C
import os

import numpy as np

from . import _ffi
from ._ffi import (PtkDenoiseTask, PtkRenderJob, PtkSceneDesc, PtkSettings, PtkStats)  # noqa: F401

_host = None
_dev = None
_dev_initialised = None


class PtkError(RuntimeError):
    pass


def host_lib():
    global _host
    if _host is None:
        _host = _ffi.load_host_library()
    return _host


def device_lib():
    global _dev
    if _dev is None:
        _dev = _ffi.load_device_library()
    return _dev


def _check_host(rc):
    if rc != 0:
        raise PtkError(host_lib().ptk_host_last_error().decode(errors="replace"))


def _check_dev(rc):
    if rc != 0:
        raise PtkError(device_lib().ptk_last_error().decode(errors="replace"))


def _dp(a):
    """Pointer to a float64, C-contiguous array (the layout every buffer argument of the C ABI has)."""
    if a is None:
        return None
    if not isinstance(a, np.ndarray) or a.dtype != np.float64 or not a.flags["C_CONTIGUOUS"]:
        raise TypeError("expected a C-contiguous numpy array of float64")
    return a.ctypes.data_as(_ffi.dptr)


def _u64p(a):
    if a is None:
        return None
    if not isinstance(a, np.ndarray) or a.dtype != np.uint64 or not a.flags["C_CONTIGUOUS"]:
        raise TypeError("expected a C-contiguous numpy array of uint64")
    return a.ctypes.data_as(_ffi.u64ptr)


# ------------------------------------------------------------------------------------------
# host side
# ------------------------------------------------------------------------------------------
def default_settings():
    s = PtkSettings()
    host_lib().ptk_host_default_settings(C.byref(s))
    return s


def load_settings(path):
    """settings.json -> PtkSettings (reference defaults for absent keys)."""
    s = PtkSettings()
    _check_host(host_lib().ptk_host_load_settings(os.fsencode(path), C.byref(s)))
    return s


class HostScene:
    """A parsed scene.json: owns the flat PtkSceneDesc the device library and the checkers take."""

    def __init__(self, handle):
        self._handle = handle
        self.desc = host_lib().ptk_host_scene_desc(handle).contents

    @classmethod
    def load(cls, scene_json_path, settings, asset_root=None):
        h = C.c_void_p()
        root = None if asset_root is None else os.fsencode(asset_root)
        _check_host(host_lib().ptk_host_load_scene(os.fsencode(scene_json_path), C.byref(settings), root, C.byref(h)))
        return cls(h)

    @classmethod
    def from_text(cls, text, settings, asset_root=None):
        h = C.c_void_p()
        root = None if asset_root is None else os.fsencode(asset_root)
        _check_host(host_lib().ptk_host_load_scene_text(text.encode(), C.byref(settings), root, C.byref(h)))
        return cls(h)

    @property
    def settings(self):
        return self.desc.settings

    @property
    def object_names(self):
        return [host_lib().ptk_host_object_name(self._handle, i).decode() for i in range(self.desc.n_objects)]

    def close(self):
        if self._handle:
            host_lib().ptk_host_free_scene(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def load_scene_directory(scene_directory, asset_root=None, **overrides):
    """`<scene_directory>/settings.json` + `scene.json`, as the reference's main() reads them.
    Keyword overrides replace settings fields (e.g. width=64, samples_per_pixel=4)."""
    settings = load_settings(os.path.join(scene_directory, "settings.json"))
    for key, value in overrides.items():
        if not hasattr(settings, key):
            raise AttributeError(f"unknown setting {key!r}")
        setattr(settings, key, int(value))
    return HostScene.load(os.path.join(scene_directory, "scene.json"), settings, asset_root or scene_directory)


def load_denoising(path, settings):
    tasks = C.POINTER(PtkDenoiseTask)()
    n = C.c_int32()
    _check_host(host_lib().ptk_host_load_denoising(os.fsencode(path), C.byref(settings), C.byref(tasks), C.byref(n)))
    out = (PtkDenoiseTask * n.value)()
    for i in range(n.value):
        out[i] = tasks[i]
    if n.value:
        C.CDLL(None).free(tasks)
    return out


def make_denoise_tasks(pipeline):
    """[{'mode': 'median', 'kernel_size': 3, 'threshold': .5}, {'mode': 'atrous', ...}] -> task array."""
    out = (PtkDenoiseTask * len(pipeline))()
    for i, t in enumerate(pipeline):
        if t["mode"] == "atrous":
            out[i].mode = _ffi.PTK_DENOISE_ATROUS
            out[i].iterations = int(t["iterations"])
            out[i].sigma_rt, out[i].sigma_x, out[i].sigma_n = float(t["sigma_rt"]), float(t["sigma_x"]), float(t["sigma_n"])
        elif t["mode"] == "median":
            out[i].mode = _ffi.PTK_DENOISE_MEDIAN
            out[i].kernel_size = int(t["kernel_size"])
            out[i].threshold = float(t["threshold"])
        else:
            raise ValueError(f"'{t['mode']}' is not a valid denoising mode.")
    return out


def write_map(path, data):
    """Writes a `.map` texture from an [H, W] or [H, W, C] array (python_utils/get_map.py:34-62)."""
    a = np.ascontiguousarray(data, dtype=np.float64)
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, d = a.shape
    _check_host(host_lib().ptk_host_write_map(os.fsencode(path), w, h, d, _dp(a)))


def make_job(start_idx, number_of_pixels, number_of_samples, sample_offset=0, seed=0):
    return PtkRenderJob(int(start_idx), int(number_of_pixels), int(number_of_samples), int(sample_offset), int(seed))


# ------------------------------------------------------------------------------------------
# device side
# ------------------------------------------------------------------------------------------
def init(device=0):
    """Selects the GPU and creates the library's stream.  Raises when there is no usable GPU."""
    global _dev_initialised
    if _dev_initialised != device:
        _check_dev(device_lib().ptk_init(int(device)))
        _dev_initialised = device


def sync():
    _check_dev(device_lib().ptk_sync())


def set_option(name, value):
    """Tuning knobs of the device library (include/ptk.h, ptk_set_option); results never depend on them."""
    _check_dev(device_lib().ptk_set_option(name.encode(), int(value)))


def set_profiling(enabled):
    """HIP-event timing of every kernel family into PtkStats.*_ms (costs a few microseconds per launch)."""
    _check_dev(device_lib().ptk_set_profiling(1 if enabled else 0))


class Film:
    """Device-resident accumulation buffers: three float64 [H, W, 3] means + per-pixel counters."""

    def __init__(self, width, height, device_buffers=None):
        """`device_buffers`: optional (image, position, normal, counters) device addresses owned by the
        caller (e.g. torch tensors' data_ptr()) to accumulate into instead of library-owned memory."""
        self.width, self.height = int(width), int(height)
        h = C.c_void_p()
        if device_buffers is None:
            _check_dev(device_lib().ptk_film_create(self.width, self.height, C.byref(h)))
        else:
            _check_dev(device_lib().ptk_film_wrap(self.width, self.height, *[C.c_void_p(int(p)) for p in device_buffers],
                                                  C.byref(h)))
        self._handle = h

    def clear(self):
        _check_dev(device_lib().ptk_film_clear(self._handle))

    def upload(self, image=None, position=None, normal=None, counters=None):
        n = self.width * self.height
        for a, size in ((image, 3 * n), (position, 3 * n), (normal, 3 * n), (counters, n)):
            if a is not None and a.size != size:
                raise ValueError("buffer size does not match the film")
        _check_dev(device_lib().ptk_film_upload(self._handle, _dp(image), _dp(position), _dp(normal), _u64p(counters)))

    def download(self):
        shape = (self.height, self.width, 3)
        image, position, normal = np.empty(shape), np.empty(shape), np.empty(shape)
        counters = np.empty((self.height, self.width), dtype=np.uint64)
        _check_dev(device_lib().ptk_film_download(self._handle, _dp(image), _dp(position), _dp(normal),
                                                  counters.ctypes.data_as(_ffi.u64ptr)))
        return image, position, normal, counters

    def merge(self, other):
        """Folds `other` (a film of the same size, on any device) into this film: the sample-count-weighted mean of the
        two running means per pixel (ptk_film_merge); `other` is left as it is."""
        _check_dev(device_lib().ptk_film_merge(self._handle, other._handle))

    def device_pointers(self):
        p = [C.c_void_p() for _ in range(4)]
        _check_dev(device_lib().ptk_film_device_pointers(self._handle, *[C.byref(x) for x in p]))
        return [x.value for x in p]

    def close(self):
        if self._handle:
            device_lib().ptk_film_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceScene:
    """A scene resident on the GPU (ptk_scene_create)."""

    def __init__(self, host_scene, max_paths_in_flight=0):
        init(_dev_initialised if _dev_initialised is not None else 0)
        # a copy: the description belongs to the host scene, which may be closed while this scene lives on
        self.settings = PtkSettings.from_buffer_copy(host_scene.desc.settings)
        h = C.c_void_p()
        _check_dev(device_lib().ptk_scene_create(C.byref(host_scene.desc), int(max_paths_in_flight), C.byref(h)))
        self._handle = h

    @property
    def n_pixels(self):
        return self.settings.width * self.settings.height

    def clone(self):
        """A copy of this scene on the current device, made device to device without building anything again
        (ptk_scene_clone)."""
        other = DeviceScene.__new__(DeviceScene)
        other.settings = PtkSettings.from_buffer_copy(self.settings)
        h = C.c_void_p()
        _check_dev(device_lib().ptk_scene_clone(self._handle, C.byref(h)))
        other._handle = h
        return other

    def trace_primary(self, job):
        n = job.number_of_pixels
        obj = np.empty(n, dtype=np.int32)
        prim = np.empty(n, dtype=np.int32)
        dist = np.empty(n)
        pos = np.empty((n, 3))
        nrm = np.empty((n, 3))
        _check_dev(device_lib().ptk_trace_primary(self._handle, C.byref(job), obj.ctypes.data_as(_ffi.i32ptr),
                                                  prim.ctypes.data_as(_ffi.i32ptr), _dp(dist), _dp(pos), _dp(nrm)))
        return obj, prim, dist, pos, nrm

    def trace_samples(self, job):
        shape = (job.number_of_pixels, job.number_of_samples, 3)
        color, pos, nrm = np.empty(shape), np.empty(shape), np.empty(shape)
        _check_dev(device_lib().ptk_trace_samples(self._handle, C.byref(job), _dp(color), _dp(pos), _dp(nrm)))
        return color, pos, nrm

    def render(self, film, job):
        _check_dev(device_lib().ptk_render(self._handle, film._handle, C.byref(job)))

    def render_bands(self, film, job, band_pixels, band_stride):
        """render() for a job made of runs of `band_pixels` pixels that begin `band_stride` pixels apart (the
        interleaved row bands of one GPU, see distributed.band_job); job.number_of_pixels counts all of them."""
        _check_dev(device_lib().ptk_render_bands(self._handle, film._handle, C.byref(job), int(band_pixels), int(band_stride)))

    def render_host(self, job, image, position, normal, counters):
        n = self.n_pixels
        if image.size != 3 * n or position.size != 3 * n or normal.size != 3 * n or counters.size != n:
            raise ValueError("buffer size does not match the scene's image")
        _check_dev(device_lib().ptk_render_host(self._handle, C.byref(job), _dp(image), _dp(position), _dp(normal),
                                                _u64p(counters)))

    def stats(self):
        s = PtkStats()
        _check_dev(device_lib().ptk_get_stats(self._handle, C.byref(s)))
        return s

    def build_info(self):
        """{'bvh_build_ms', 'bvh_build_wall_ms', 'meshes_built_on_gpu', 'meshes_built_on_host', 'bvh_nodes'} of this
        scene's mesh hierarchies (bvh_build_ms: device time of the build kernels for builds on the GPU; the wall figure
        includes their allocations, copies and waits)."""
        ms, gpu, host, nodes, wall = C.c_double(), C.c_int32(), C.c_int32(), C.c_int64(), C.c_double()
        _check_dev(device_lib().ptk_scene_build_info(self._handle, C.byref(ms), C.byref(gpu), C.byref(host), C.byref(nodes)))
        _check_dev(device_lib().ptk_scene_build_wall_ms(self._handle, C.byref(wall)))
        return {"bvh_build_ms": ms.value, "bvh_build_wall_ms": wall.value, "meshes_built_on_gpu": gpu.value,
                "meshes_built_on_host": host.value, "bvh_nodes": nodes.value}

    def bounce_kernel(self):
        """Profiler name of the k_bounce instantiation this scene uses."""
        return device_lib().ptk_scene_bounce_kernel(self._handle).decode()

    def reset_stats(self):
        _check_dev(device_lib().ptk_reset_stats(self._handle))

    def close(self):
        if self._handle:
            device_lib().ptk_scene_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def denoise_film(tasks, film, scratch_film):
    """Applies a denoising pipeline to the image of a device-resident film, in place and asynchronously on the
    library's stream; `scratch_film` (same size) lends its image buffer as scratch space."""
    img, pos, nrm, _ = film.device_pointers()
    scratch = scratch_film.device_pointers()[0]
    arr = (PtkDenoiseTask * len(tasks))(*tasks)
    _check_dev(device_lib().ptk_denoise_device(arr, len(tasks), film.width, film.height, img, pos, nrm, scratch))


def denoise_film_static(tasks, film):
    """denoise_film without a scratch film and without any device allocation: the library's static scratch
    buffer is used (images of at most 2,097,152 pixels; see ptk_denoise_device_static in include/ptk.h)."""
    img, pos, nrm, _ = film.device_pointers()
    arr = (PtkDenoiseTask * len(tasks))(*tasks)
    _check_dev(device_lib().ptk_denoise_device_static(arr, len(tasks), film.width, film.height, img, pos, nrm))


def denoise(tasks, image, position, normal):
    """Applies a denoising pipeline on the GPU; returns the filtered copy of `image`."""
    init(_dev_initialised if _dev_initialised is not None else 0)
    out = np.array(image, dtype=np.float64, order="C", copy=True)
    pos = np.ascontiguousarray(position, dtype=np.float64)
    nrm = np.ascontiguousarray(normal, dtype=np.float64)
    h, w = out.shape[0], out.shape[1]
    _check_dev(device_lib().ptk_denoise(tasks, len(tasks), w, h, _dp(out), _dp(pos), _dp(nrm)))
    return out
