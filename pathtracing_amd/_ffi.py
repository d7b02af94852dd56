"""ctypes mirror of include/ptk.h and include/ptk_host.h (struct layouts and prototypes)."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))

c_double3 = C.c_double * 3


class PtkMap(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("dim", C.c_int32), ("_pad", C.c_int32),
                ("offset", C.c_int64)]


class PtkMedium(C.Structure):
    _fields_ = [("type", C.c_int32), ("_pad", C.c_int32), ("scattering", c_double3), ("absorption", c_double3),
                ("emission", c_double3), ("refractive_index", C.c_double)]


class PtkMaterial(C.Structure):
    _fields_ = [("type", C.c_int32), ("is_light_source", C.c_int32), ("is_dielectric", C.c_int32),
                ("albedo_map", C.c_int32), ("emission_color_map", C.c_int32), ("light_intensity_map", C.c_int32),
                ("roughness_map", C.c_int32), ("transparency_map", C.c_int32), ("internal_medium", C.c_int32),
                ("external_medium", C.c_int32), ("surface_refractive_index", C.c_double),
                ("extinction_coefficient", C.c_double)]


class PtkObject(C.Structure):
    _fields_ = [("type", C.c_int32), ("material", C.c_int32), ("position", c_double3), ("v1", c_double3),
                ("v2", c_double3), ("L1", C.c_double), ("L2", C.c_double), ("radius", C.c_double),
                ("first_triangle", C.c_int64), ("n_triangles", C.c_int64)]


class PtkCamera(C.Structure):
    _fields_ = [("position", c_double3), ("orientation_x", C.c_double), ("orientation_y", C.c_double),
                ("orientation_z", C.c_double), ("camera_width", C.c_double), ("focal_length", C.c_double),
                ("aperture_size", C.c_double), ("focus_distance", C.c_double), ("dof_mode", C.c_int32),
                ("_pad", C.c_int32)]


class PtkSettings(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("samples_per_pixel", C.c_int32),
                ("samples_per_iteration", C.c_int32), ("max_recursion_depth", C.c_int32),
                ("min_recursion_steps", C.c_int32), ("number_of_threads", C.c_int32),
                ("use_gamma_correction", C.c_int32), ("bvh_leaf_size", C.c_int32), ("bvh_n_axis_splits", C.c_int32),
                ("enable_next_event_estimation", C.c_int32), ("enable_anti_aliasing", C.c_int32)]


class PtkSceneDesc(C.Structure):
    _fields_ = [("settings", PtkSettings), ("camera", PtkCamera), ("n_objects", C.c_int32),
                ("n_materials", C.c_int32), ("n_media", C.c_int32), ("n_maps", C.c_int32),
                ("objects", C.POINTER(PtkObject)), ("materials", C.POINTER(PtkMaterial)),
                ("media", C.POINTER(PtkMedium)), ("maps", C.POINTER(PtkMap)), ("map_data", C.POINTER(C.c_double)),
                ("n_map_values", C.c_int64), ("n_triangles", C.c_int64), ("tri_vertices", C.POINTER(C.c_double)),
                ("tri_uv", C.POINTER(C.c_double)), ("tri_normals", C.POINTER(C.c_double)),
                ("tri_flags", C.POINTER(C.c_uint8)), ("background_medium", C.c_int32), ("_pad", C.c_int32),
                ("background_color", c_double3)]


class PtkRenderJob(C.Structure):
    _fields_ = [("start_idx", C.c_int64), ("number_of_pixels", C.c_int64), ("number_of_samples", C.c_int32),
                ("sample_offset", C.c_int32), ("seed", C.c_uint64)]


class PtkStats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("extension_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("vertices", C.c_uint64), ("bounce_launch_rounds", C.c_uint64), ("algorithmic_bytes", C.c_uint64),
                ("bounce_ms", C.c_double), ("bin_ms", C.c_double), ("other_ms", C.c_double),
                ("traverse_ms", C.c_double), ("bounce_launches", C.c_uint64), ("bin_launches", C.c_uint64),
                ("traverse_launches", C.c_uint64), ("bounce_bytes", C.c_uint64), ("bin_bytes", C.c_uint64),
                ("traverse_bytes", C.c_uint64), ("class_vertices", C.c_uint64 * 8), ("traversed_rays", C.c_uint64),
                ("nodes_visited", C.c_uint64), ("triangles_tested", C.c_uint64), ("stack_overflows", C.c_uint64)]


class PtkDenoiseTask(C.Structure):
    _fields_ = [("mode", C.c_int32), ("iterations", C.c_int32), ("sigma_rt", C.c_double), ("sigma_x", C.c_double),
                ("sigma_n", C.c_double), ("kernel_size", C.c_int32), ("_pad", C.c_int32), ("threshold", C.c_double)]


PTK_OBJ_SPHERE, PTK_OBJ_PLANE, PTK_OBJ_RECTANGLE, PTK_OBJ_TRIANGLE, PTK_OBJ_MESH = range(5)
(PTK_MAT_DIFFUSE, PTK_MAT_REFLECTIVE, PTK_MAT_TRANSPARENT, PTK_MAT_GLOSSY, PTK_MAT_METALLIC_MICROFACET,
 PTK_MAT_REFLECTIVE_MICROFACET, PTK_MAT_TRANSPARENT_MICROFACET) = range(7)
PTK_DENOISE_ATROUS, PTK_DENOISE_MEDIAN = 0, 1
PTK_TRI_SMOOTH, PTK_TRI_HAS_UV = 1, 2

dptr = C.POINTER(C.c_double)
i32ptr = C.POINTER(C.c_int32)
u64ptr = C.POINTER(C.c_uint64)

# name -> (restype, argtypes); also the list of symbols include/ptk_host.h declares
HOST_PROTOTYPES = {
    "ptk_host_last_error": (C.c_char_p, []),
    "ptk_host_default_settings": (None, [C.POINTER(PtkSettings)]),
    "ptk_host_load_settings": (C.c_int, [C.c_char_p, C.POINTER(PtkSettings)]),
    "ptk_host_load_scene": (C.c_int, [C.c_char_p, C.POINTER(PtkSettings), C.c_char_p, C.POINTER(C.c_void_p)]),
    "ptk_host_load_scene_text": (C.c_int, [C.c_char_p, C.POINTER(PtkSettings), C.c_char_p, C.POINTER(C.c_void_p)]),
    "ptk_host_scene_desc": (C.POINTER(PtkSceneDesc), [C.c_void_p]),
    "ptk_host_object_name": (C.c_char_p, [C.c_void_p, C.c_int32]),
    "ptk_host_free_scene": (None, [C.c_void_p]),
    "ptk_host_load_denoising": (C.c_int, [C.c_char_p, C.POINTER(PtkSettings), C.POINTER(C.POINTER(PtkDenoiseTask)),
                                          C.POINTER(C.c_int32)]),
    "ptk_host_map_buffer": (C.c_void_p, [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_int)]),
    "ptk_host_unmap_buffer": (None, [C.c_void_p, C.c_size_t, C.c_int]),
    "ptk_host_write_map": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, dptr]),
}

# symbols include/ptk.h declares
DEVICE_PROTOTYPES = {
    "ptk_init": (C.c_int, [C.c_int]),
    "ptk_last_error": (C.c_char_p, []),
    "ptk_sync": (C.c_int, []),
    "ptk_stream": (C.c_void_p, []),
    "ptk_scene_create": (C.c_int, [C.POINTER(PtkSceneDesc), C.c_int64, C.POINTER(C.c_void_p)]),
    "ptk_scene_destroy": (None, [C.c_void_p]),
    "ptk_trace_primary": (C.c_int, [C.c_void_p, C.POINTER(PtkRenderJob), i32ptr, i32ptr, dptr, dptr, dptr]),
    "ptk_trace_samples": (C.c_int, [C.c_void_p, C.POINTER(PtkRenderJob), dptr, dptr, dptr]),
    "ptk_film_create": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "ptk_film_wrap": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.POINTER(C.c_void_p)]),
    "ptk_film_destroy": (None, [C.c_void_p]),
    "ptk_film_clear": (C.c_int, [C.c_void_p]),
    "ptk_film_upload": (C.c_int, [C.c_void_p, dptr, dptr, dptr, u64ptr]),
    "ptk_film_download": (C.c_int, [C.c_void_p, dptr, dptr, dptr, u64ptr]),
    "ptk_film_device_pointers": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                           C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "ptk_film_merge": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ptk_scene_clone": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "ptk_render": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(PtkRenderJob)]),
    "ptk_render_bands": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(PtkRenderJob), C.c_int64, C.c_int64]),
    "ptk_render_host": (C.c_int, [C.c_void_p, C.POINTER(PtkRenderJob), dptr, dptr, dptr, u64ptr]),
    "ptk_scene_build_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int64)]),
    "ptk_scene_build_wall_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "ptk_get_stats": (C.c_int, [C.c_void_p, C.POINTER(PtkStats)]),
    "ptk_scene_bounce_kernel": (C.c_char_p, [C.c_void_p]),
    "ptk_reset_stats": (C.c_int, [C.c_void_p]),
    "ptk_set_profiling": (C.c_int, [C.c_int]),
    "ptk_set_option": (C.c_int, [C.c_char_p, C.c_int64]),
    "ptk_selftest_record_staging": (C.c_int, [C.c_int64, C.POINTER(C.c_int64)]),
    "ptk_denoise": (C.c_int, [C.POINTER(PtkDenoiseTask), C.c_int32, C.c_int32, C.c_int32, dptr, dptr, dptr]),
    "ptk_denoise_device": (C.c_int, [C.POINTER(PtkDenoiseTask), C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    "ptk_denoise_device_static": (C.c_int, [C.POINTER(PtkDenoiseTask), C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                            C.c_void_p, C.c_void_p]),
}


def bind(lib, prototypes):
    for name, (restype, argtypes) in prototypes.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


def load_host_library():
    path = os.path.join(HERE, "libptkhost.so")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
    return bind(C.CDLL(path), HOST_PROTOTYPES)


def load_device_library():
    # PTK_DEVICE_LIBRARY selects another build of the same library (kernel tuning experiments)
    path = os.environ.get("PTK_DEVICE_LIBRARY") or os.path.join(HERE, "libptk.so")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: the HIP library must be built (there is no CPU fallback); "
                           "run `python -c 'import __graft_entry__ as g; g.build()'`")
    return bind(C.CDLL(path), DEVICE_PROTOTYPES)
