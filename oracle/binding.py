"""TEST INFRASTRUCTURE: ctypes bindings of the CPU checker (liboracle.so) and, in the build
container only, of the reference harness (_ref/libref.so).  Imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg -- never by pathtracing_amd/."""
import ctypes as C
import os

import numpy as np

from pathtracing_amd import _ffi
from pathtracing_amd._ffi import PtkDenoiseTask, PtkRenderJob, PtkSceneDesc, dptr, i32ptr, u64ptr

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_RNG_COUNTER, ORACLE_RNG_REFERENCE_STREAMS = 0, 1


class OracleRng(C.Structure):
    _fields_ = [("mode", C.c_int32), ("uniform_seed", C.c_uint32), ("normal_seed", C.c_uint32),
                ("rand_seed", C.c_uint32)]


class OracleStats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("extension_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("surface_vertices", C.c_uint64), ("scatter_vertices", C.c_uint64),
                ("bvh_nodes_visited", C.c_uint64), ("triangles_tested", C.c_uint64),
                ("vertices_by_material", C.c_uint64 * 7)]


d3 = C.c_double * 3

ORACLE_PROTOTYPES = {
    "oracle_last_error": (C.c_char_p, []),
    "oracle_scene_create": (C.c_int, [C.POINTER(PtkSceneDesc), C.POINTER(C.c_void_p)]),
    "oracle_scene_destroy": (None, [C.c_void_p]),
    "oracle_mesh_face_index": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_int64)]),
    "oracle_trace_primary": (C.c_int, [C.c_void_p, C.POINTER(PtkRenderJob), C.POINTER(OracleRng), i32ptr, i32ptr, dptr,
                                       dptr, dptr]),
    "oracle_trace_samples": (C.c_int, [C.c_void_p, C.POINTER(PtkRenderJob), C.POINTER(OracleRng), dptr, dptr, dptr]),
    "oracle_render": (C.c_int, [C.c_void_p, C.POINTER(PtkRenderJob), C.POINTER(OracleRng), C.c_int32, dptr, dptr, dptr,
                                u64ptr]),
    "oracle_get_stats": (C.c_int, [C.c_void_p, C.POINTER(OracleStats)]),
    "oracle_denoise": (C.c_int, [C.POINTER(PtkDenoiseTask), C.c_int32, C.c_int32, C.c_int32, dptr, dptr, dptr]),
    "oracle_rng_key": (C.c_uint64, [C.c_uint64, C.c_uint32, C.c_uint32]),
    "oracle_rng_uniform": (C.c_double, [C.c_uint64, C.c_uint32]),
    "oracle_rng_fork": (C.c_uint64, [C.c_uint64, C.c_uint32]),
    "oracle_fresnel_dielectric": (C.c_double, [C.c_double] * 3),
    "oracle_fresnel_conductor": (C.c_double, [C.c_double] * 5),
    "oracle_schlick_fresnel": (C.c_double, [C.c_double] * 3),
    "oracle_beckmann_D": (C.c_double, [d3, d3, C.c_double]),
    "oracle_smith_G1": (C.c_double, [d3, d3, d3, C.c_double]),
    "oracle_rotate": (None, [d3, C.c_double, C.c_double, C.c_double, d3]),
    "oracle_refract": (None, [d3, d3, C.c_double, d3]),
    "oracle_solve_quadratic": (C.c_int, [C.c_double, C.c_double, dptr]),
}

REF_PROTOTYPES = {
    "ref_last_error": (C.c_char_p, []),
    "ref_scene_create": (C.c_int, [C.POINTER(PtkSceneDesc), C.POINTER(C.c_void_p)]),
    "ref_scene_destroy": (None, [C.c_void_p]),
    "ref_seed": (None, [C.c_uint32, C.c_uint32, C.c_uint32]),
    "ref_trace_primary": (C.c_int, [C.c_void_p, C.POINTER(PtkRenderJob), i32ptr, i32ptr, dptr, dptr, dptr]),
    "ref_trace_samples": (C.c_int, [C.c_void_p, C.POINTER(PtkRenderJob), dptr, dptr, dptr]),
    "ref_render": (C.c_int, [C.c_void_p, C.c_int32, dptr, dptr, dptr]),
    "ref_denoise": (C.c_int, [C.POINTER(PtkDenoiseTask), C.c_int32, C.c_int32, C.c_int32, dptr, dptr, dptr]),
    "ref_fresnel_dielectric": (C.c_double, [C.c_double] * 3),
    "ref_fresnel_conductor": (C.c_double, [C.c_double] * 5),
    "ref_schlick_fresnel": (C.c_double, [C.c_double] * 3),
    "ref_beckmann_D": (C.c_double, [d3, d3, C.c_double]),
    "ref_smith_G1": (C.c_double, [d3, d3, d3, C.c_double]),
    "ref_rotate": (None, [d3, C.c_double, C.c_double, C.c_double, d3]),
    "ref_refract": (None, [d3, d3, C.c_double, d3]),
    "ref_solve_quadratic": (C.c_int, [C.c_double, C.c_double, dptr]),
}

_oracle = None
_ref = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle oracle`")
        _oracle = _ffi.bind(C.CDLL(path), ORACLE_PROTOTYPES)
    return _oracle


def ref_available():
    return os.path.exists(os.path.join(HERE, "_ref", "libref.so"))


def ref_lib():
    global _ref
    if _ref is None:
        _ref = _ffi.bind(C.CDLL(os.path.join(HERE, "_ref", "libref.so")), REF_PROTOTYPES)
    return _ref


def _dp(a):
    return None if a is None else a.ctypes.data_as(dptr)


def counter_rng():
    return OracleRng(ORACLE_RNG_COUNTER, 0, 0, 0)


def stream_rng(uniform_seed, normal_seed, rand_seed):
    return OracleRng(ORACLE_RNG_REFERENCE_STREAMS, uniform_seed, normal_seed, rand_seed)


class _SceneBase:
    prefix = None
    lib = None

    def _call(self, name, *args):
        rc = getattr(self.lib, f"{self.prefix}_{name}")(*args)
        if rc != 0:
            raise RuntimeError(getattr(self.lib, f"{self.prefix}_last_error")().decode())

    def __init__(self, host_scene):
        self.host_scene = host_scene  # keeps the description's arrays alive
        self.settings = host_scene.desc.settings
        h = C.c_void_p()
        self._call("scene_create", C.byref(host_scene.desc), C.byref(h))
        self._handle = h

    def close(self):
        if self._handle:
            getattr(self.lib, f"{self.prefix}_scene_destroy")(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class OracleScene(_SceneBase):
    prefix = "oracle"

    def __init__(self, host_scene):
        self.lib = oracle_lib()
        super().__init__(host_scene)

    def trace_primary(self, job, rng=None):
        rng = rng or counter_rng()
        n = job.number_of_pixels
        obj, prim = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.int32)
        dist, pos, nrm = np.empty(n), np.empty((n, 3)), np.empty((n, 3))
        self._call("trace_primary", self._handle, C.byref(job), C.byref(rng), obj.ctypes.data_as(i32ptr),
                   prim.ctypes.data_as(i32ptr), _dp(dist), _dp(pos), _dp(nrm))
        return obj, prim, dist, pos, nrm

    def trace_samples(self, job, rng=None):
        rng = rng or counter_rng()
        shape = (job.number_of_pixels, job.number_of_samples, 3)
        color, pos, nrm = np.empty(shape), np.empty(shape), np.empty(shape)
        self._call("trace_samples", self._handle, C.byref(job), C.byref(rng), _dp(color), _dp(pos), _dp(nrm))
        return color, pos, nrm

    def render(self, job, image, position, normal, counters, rng=None, threads=1):
        rng = rng or counter_rng()
        self._call("render", self._handle, C.byref(job), C.byref(rng), int(threads), _dp(image), _dp(position),
                   _dp(normal), counters.ctypes.data_as(u64ptr))

    def stats(self):
        s = OracleStats()
        self._call("get_stats", self._handle, C.byref(s))
        return s


class RefScene(_SceneBase):
    """The reference's own classes, built from the same description (build container only)."""
    prefix = "ref"

    def __init__(self, host_scene):
        self.lib = ref_lib()
        super().__init__(host_scene)

    def seed(self, uniform_seed, normal_seed, rand_seed):
        self.lib.ref_seed(uniform_seed, normal_seed, rand_seed)

    def trace_primary(self, job):
        n = job.number_of_pixels
        obj, prim = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.int32)
        dist, pos, nrm = np.empty(n), np.empty((n, 3)), np.empty((n, 3))
        self._call("trace_primary", self._handle, C.byref(job), obj.ctypes.data_as(i32ptr),
                   prim.ctypes.data_as(i32ptr), _dp(dist), _dp(pos), _dp(nrm))
        return obj, prim, dist, pos, nrm

    def trace_samples(self, job):
        shape = (job.number_of_pixels, job.number_of_samples, 3)
        color, pos, nrm = np.empty(shape), np.empty(shape), np.empty(shape)
        self._call("trace_samples", self._handle, C.byref(job), _dp(color), _dp(pos), _dp(nrm))
        return color, pos, nrm

    def render(self, threads=1):
        shape = (self.settings.height, self.settings.width, 3)
        image, pos, nrm = np.zeros(shape), np.zeros(shape), np.zeros(shape)
        self._call("render", self._handle, int(threads), _dp(image), _dp(pos), _dp(nrm))
        return image, pos, nrm


def oracle_denoise(tasks, image, position, normal):
    out = np.array(image, dtype=np.float64, order="C", copy=True)
    pos = np.ascontiguousarray(position, dtype=np.float64)
    nrm = np.ascontiguousarray(normal, dtype=np.float64)
    rc = oracle_lib().oracle_denoise(tasks, len(tasks), out.shape[1], out.shape[0], _dp(out), _dp(pos), _dp(nrm))
    if rc != 0:
        raise RuntimeError(oracle_lib().oracle_last_error().decode())
    return out


def ref_denoise(tasks, image, position, normal):
    out = np.array(image, dtype=np.float64, order="C", copy=True)
    pos = np.array(position, dtype=np.float64, order="C", copy=True)
    nrm = np.array(normal, dtype=np.float64, order="C", copy=True)
    ref_lib().ref_denoise(tasks, len(tasks), out.shape[1], out.shape[0], _dp(out), _dp(pos), _dp(nrm))
    return out
