/*
 * ptk.h -- C ABI of the MI355X (gfx950) path-tracing hot path.
 *
 * The reference renderer (JoarGunnarsson/Pathtracing) has no plugin or FFI seam; the
 * boundary this library replaces is its internal C++ call boundary between the scene
 * loader and the render/denoise loops:
 *
 *   raytrace_section(job, scene, buffers, ctx)        app/pathtracer.cpp:237-275
 *   compute_pixel_color / raytrace                     app/pathtracer.cpp:39-201
 *   find_closest_hit                                   src/objects.cpp:414-442
 *   denoise(PixelBuffers&, vector<DenoisingTask>)      src/denoise.cpp:203-212
 *
 * The polymorphic Scene graph (include/scene.h:48-55) is replaced by the flat,
 * pointer-and-size PtkSceneDesc below.  Everything is plain C: no C++ types, no torch
 * types.  All functions return 0 on success, a negative PtkStatus on failure; the
 * message is available from ptk_last_error() (thread local).  Nothing throws across
 * the boundary.
 *
 * Random numbers.  The reference draws from three process-global sequential generators
 * (src/utils.cpp:11-30, 120-121), which cannot be reproduced by a parallel machine.
 * This ABI fixes a counter-based generator instead, so that any implementation (the
 * GPU kernels, a CPU checker) produces the same numbers for the same logical draw:
 *
 *   mix64(z):  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9; z ^= z >> 27;
 *              z *= 0x94D049BB133111EB; z ^= z >> 31;
 *   key(seed, pixel, sample) = mix64(mix64(seed + G) ^ ((uint64)pixel << 32 | sample))
 *   bits(key, n)  = mix64(key + (n + 1) * G)            G = 0x9E3779B97F4A7C15
 *   uniform(key,n)= (bits >> 11) * 2^-53                in [0, 1)
 *   normal        = sqrt(-2 ln(1 - u(n))) * cos(2 pi u(n+1))      consumes two counters
 *   fork(key, n)  = mix64(key ^ mix64(n + F))           F = 0xD1B54A32D192ED03; consumes one
 *
 * `pixel` is the linear pixel index row*W + x, `sample` the global sample index of that
 * pixel (job.sample_offset + i).  A path consumes counters 0,1,2,... in the order in
 * which the reference algorithm calls its generators (AA jitter, lens, then per depth:
 * free-flight distance, light pick, light point, eval, [visibility march on a forked
 * key], BRDF sample, virtual-surface test, Russian roulette).  Results therefore do not
 * depend on batch size, launch geometry or the number of GPUs.
 */
#ifndef PTK_H
#define PTK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum PtkStatus {
    PTK_OK = 0,
    PTK_ERR_INVALID = -1,  /* bad argument / malformed scene description */
    PTK_ERR_DEVICE = -2,   /* HIP runtime error or no usable GPU */
    PTK_ERR_MEMORY = -3,   /* allocation failure or size over a stated limit */
    PTK_ERR_IO = -4,       /* file missing / unreadable (host loader) */
    PTK_ERR_FORMAT = -5    /* JSON / OBJ / .map content error (host loader) */
} PtkStatus;

/* ---- scene description -------------------------------------------------------- */

/* Object subtypes of scene.json (src/scene.cpp:358-419). */
enum { PTK_OBJ_SPHERE = 0, PTK_OBJ_PLANE = 1, PTK_OBJ_RECTANGLE = 2, PTK_OBJ_TRIANGLE = 3, PTK_OBJ_MESH = 4 };
/* Material subtypes (src/scene.cpp:321-344). */
enum {
    PTK_MAT_DIFFUSE = 0,
    PTK_MAT_REFLECTIVE = 1,
    PTK_MAT_TRANSPARENT = 2,
    PTK_MAT_GLOSSY = 3,
    PTK_MAT_METALLIC_MICROFACET = 4,
    PTK_MAT_REFLECTIVE_MICROFACET = 5,
    PTK_MAT_TRANSPARENT_MICROFACET = 6,
    PTK_MAT_COUNT = 7
};
/* Medium subtypes (src/scene.cpp:253-262). */
enum { PTK_MEDIUM_BEERS_LAW = 0, PTK_MEDIUM_HOMOGENEOUS_SCATTERING = 1 };
/* Camera depth-of-field modes (src/camera.cpp:39-51). */
enum { PTK_DOF_NONE = 0, PTK_DOF_CIRCLE = 1, PTK_DOF_SQUARE = 2 };

/* A ValueMap1D / ValueMap3D (include/valuemap.h): `dim` channels, row-major
 * [height][width][dim] doubles starting at map_data[offset].  3-channel data is
 * already converted sRGB->linear when use_gamma_correction is set, as the reference
 * does at load time (src/scene.cpp:218-235). */
typedef struct PtkMap {
    int32_t width, height, dim, _pad;
    int64_t offset;
} PtkMap;

/* include/medium.h.  Values are the raw JSON parameters; a Beer's-law medium ignores
 * `scattering` (src/medium.cpp:37-39). */
typedef struct PtkMedium {
    int32_t type, _pad;
    double scattering[3];
    double absorption[3];
    double emission[3];
    double refractive_index;
} PtkMedium;

/* MaterialData (include/materials.h:41-53).  Map and medium fields are indices into
 * the tables of the scene description, -1 when the JSON omits them (the reference
 * then substitutes black / 0 maps and a null medium). */
typedef struct PtkMaterial {
    int32_t type;
    int32_t is_light_source;
    int32_t is_dielectric;
    int32_t albedo_map, emission_color_map; /* 3-channel */
    int32_t light_intensity_map, roughness_map, transparency_map; /* 1-channel */
    int32_t internal_medium, external_medium;
    double surface_refractive_index;
    double extinction_coefficient;
} PtkMaterial;

/* One top-level object, in scene order.  Field use by type:
 *   sphere     position, radius
 *   plane      position, v1, v2                (un-normalised, as in JSON)
 *   rectangle  position, v1, v2, L1, L2
 *   triangle   position = p1, v1 = p2, v2 = p3
 *   mesh       first_triangle, n_triangles into the triangle arrays; smooth shading
 *              and UVs are per triangle (tri_flags).                               */
typedef struct PtkObject {
    int32_t type, material;
    double position[3];
    double v1[3];
    double v2[3];
    double L1, L2, radius;
    int64_t first_triangle, n_triangles;
} PtkObject;

enum { PTK_TRI_SMOOTH = 1, PTK_TRI_HAS_UV = 2 };

typedef struct PtkCamera {
    double position[3];
    double orientation_x, orientation_y, orientation_z; /* degrees, JSON "orientation" */
    double camera_width, focal_length;
    double aperture_size, focus_distance;
    int32_t dof_mode, _pad;
} PtkCamera;

/* settings.json keys that affect the result (src/constants.cpp:7-19). */
typedef struct PtkSettings {
    int32_t width, height;
    int32_t samples_per_pixel, samples_per_iteration;
    int32_t max_recursion_depth, min_recursion_steps;
    int32_t number_of_threads, use_gamma_correction;
    int32_t bvh_leaf_size, bvh_n_axis_splits;
    int32_t enable_next_event_estimation, enable_anti_aliasing;
} PtkSettings;

typedef struct PtkSceneDesc {
    PtkSettings settings;
    PtkCamera camera;
    int32_t n_objects, n_materials, n_media, n_maps;
    const PtkObject* objects;
    const PtkMaterial* materials;
    const PtkMedium* media;
    const PtkMap* maps;
    const double* map_data;
    int64_t n_map_values;
    /* triangles of all meshes, in OBJ face order: vertices [n][9], uv [n][6],
     * vertex normals [n][9], flags [n] */
    int64_t n_triangles;
    const double* tri_vertices;
    const double* tri_uv;
    const double* tri_normals;
    const uint8_t* tri_flags;
    int32_t background_medium, _pad;
    double background_color[3];
} PtkSceneDesc;

/* ---- device side ---------------------------------------------------------------- */

typedef struct PtkScene PtkScene; /* device-resident flattened scene + path pools */
typedef struct PtkFilm PtkFilm;   /* device-resident accumulation buffers */

/* The job unit of the reference's scheduler (Job, app/pathtracer.cpp:26-30) plus the
 * stream coordinates the reference leaves implicit. */
typedef struct PtkRenderJob {
    int64_t start_idx;         /* first linear pixel index */
    int64_t number_of_pixels;  /* contiguous pixels */
    int32_t number_of_samples; /* samples added to every pixel of the range */
    int32_t sample_offset;     /* global index of the first of those samples */
    uint64_t seed;
} PtkRenderJob;

typedef struct PtkStats {
    uint64_t paths;         /* camera samples traced */
    uint64_t extension_rays;/* closest-hit queries from the path loop */
    uint64_t shadow_rays;   /* closest-hit queries from visibility marches */
    uint64_t vertices;      /* surface + scatter vertices shaded */
    uint64_t bounce_launch_rounds;
    uint64_t algorithmic_bytes; /* path-state/queue bytes the kernels are specified to move */
    /* Kernel families: bounce = k_bounce and the tail kernel (shading, shadow ray and extension ray
     * of one path vertex), bin = the class sort between bounces, traverse = the hierarchy walks of
     * scenes with meshes (k_traverse), other = the rest (camera rays, and in scenes with meshes the
     * pass that completes the vertices which waited for k_traverse). */
    double bounce_ms, bin_ms, other_ms, traverse_ms; /* device time per family, HIP events, while profiling is on */
    uint64_t bounce_launches, bin_launches, traverse_launches;
    uint64_t bounce_bytes, bin_bytes, traverse_bytes; /* algorithmic bytes per family */
    /* Vertices shaded per class (PTK_MAT_* order, then medium scattering). */
    uint64_t class_vertices[8];
    uint64_t traversed_rays, nodes_visited, triangles_tested; /* k_traverse */
    /* Entries the hierarchy walks had to drop because their stack was full.  Always 0: the host bounds the
     * stack every hierarchy needs when it builds it; a non-zero value would mean missed triangles. */
    uint64_t stack_overflows;
} PtkStats;

/* Selects the GPU for the calling thread (hipSetDevice) and, the first time a device is
 * named, creates the library's stream on it.  Fails with PTK_ERR_DEVICE when no gfx950
 * device is usable: there is no CPU path.
 * Several devices in one process: every host thread calls ptk_init with its own device
 * and then creates its scenes and films; those remember their device, so later calls on
 * them may come from any thread (one thread at a time per scene or film).  Calls without
 * a scene or film argument (ptk_sync, ptk_stream, ptk_film_create, ptk_denoise*) act on
 * the device of the calling thread's last ptk_init, or of the process's last one for a
 * thread that never called it.  Replaces the reference's thread pool over one shared
 * scene (app/pathtracer.cpp:356-365) as the unit of host parallelism. */
int ptk_init(int device);
const char* ptk_last_error(void);
/* Blocks until all work queued by the library on the calling thread's device has finished. */
int ptk_sync(void);
/* The library's HIP stream as an opaque handle (hipStream_t). */
void* ptk_stream(void);

/* Replaces Scene construction (src/scene.cpp:546-584) as seen by the render loop:
 * copies the description to the device, builds the mesh acceleration structure
 * (replaces BVH::Node::Node, src/bvh.cpp:129-165) and sizes the path pools.
 * `max_paths_in_flight` <= 0 selects the default (3 x 2^24: three full batches); at most 2^26 is used: a job larger than a batch
 * (2^24 paths at most) is traced as batches of whole pixels, two or more at a time when it takes several (option
 * batch_lanes).  The caller may free every array of `desc` on return. */
int ptk_scene_create(const PtkSceneDesc* desc, int64_t max_paths_in_flight, PtkScene** out);
void ptk_scene_destroy(PtkScene* scene);

/* Replaces Camera::make_ray + find_closest_hit for one primary ray per pixel of the
 * range with anti-aliasing as configured (src/camera.cpp:57-70, src/objects.cpp:414-442).
 * Outputs (host pointers, may be NULL): object index in scene order (-1 = miss),
 * primitive index within its mesh in OBJ face order (0 otherwise), distance,
 * intersection point [n][3], normal facing the ray [n][3]. */
int ptk_trace_primary(PtkScene* scene, const PtkRenderJob* job, int32_t* object_index, int32_t* primitive_index,
                      double* distance, double* position, double* normal);

/* Replaces raytrace (app/pathtracer.cpp:39-176) for every (pixel, sample) of the job
 * and returns the un-averaged per-sample results to host arrays laid out
 * [pixel][sample][3]: radiance, first-surface position, first-surface normal
 * (PixelData, app/pathtracer.cpp:20-24).  Any pointer may be NULL. */
int ptk_trace_samples(PtkScene* scene, const PtkRenderJob* job, double* color, double* position, double* normal);

int ptk_film_create(int32_t width, int32_t height, PtkFilm** out);
/* A film over device buffers the caller owns (for instance tensors that a collective
 * library reduces across GPUs); ptk_film_destroy() then leaves the buffers alone. */
int ptk_film_wrap(int32_t width, int32_t height, double* d_image, double* d_position, double* d_normal,
                  uint64_t* d_counters, PtkFilm** out);
void ptk_film_destroy(PtkFilm* film);
int ptk_film_clear(PtkFilm* film);
/* Host <-> device copies of the three double[H*W*3] mean buffers and the per-pixel
 * sample counters (PixelBuffers, include/utils.h:16-20; pixel_counters,
 * app/pathtracer.cpp:357).  NULL pointers are skipped. */
int ptk_film_upload(PtkFilm* film, const double* image, const double* position, const double* normal,
                    const uint64_t* counters);
int ptk_film_download(PtkFilm* film, double* image, double* position, double* normal, uint64_t* counters);
/* Device addresses of the film's buffers, for collectives issued by the caller. */
int ptk_film_device_pointers(PtkFilm* film, double** image, double** position, double** normal,
                             uint64_t** counters);

/* Replaces raytrace_section (app/pathtracer.cpp:237-275): traces the job and folds
 * the per-pixel sample means into the film with increment_average's running-mean
 * arithmetic (app/pathtracer.cpp:232-235).  The job is cut into batches of whole pixels;
 * the call returns once the last batch has been traced (it reads that batch's ray
 * statistics back), with only the film update of the last batch still queued on the
 * stream: ptk_sync() or any ptk_film_* copy waits for it.  The film must live on the
 * scene's device. */
int ptk_render(PtkScene* scene, PtkFilm* film, const PtkRenderJob* job);

/* Folds the film `from` into the film `into` (same size; any two devices): per pixel the sample-count-weighted mean of
 * the two running means with increment_average's arithmetic (app/pathtracer.cpp:232-235), the counters added; a pixel
 * only one of the films has samples for keeps that film's bits, so films of disjoint pixels merge exactly.  `from` is
 * left as it is.  Films on different devices are copied device to device (over xGMI between the GPUs of a node), not
 * through the host.  This is the combine step of several GPUs rendering one image, for which the reference has one
 * set of buffers shared by its threads (app/pathtracer.cpp:237-275, 337-365).  Returns when the merge is done. */
int ptk_film_merge(PtkFilm* into, PtkFilm* from);

/* ptk_render for a job whose pixels are not one contiguous run but bands: runs of `band_pixels` pixels that begin
 * `band_stride` pixels apart, the first at job->start_idx; job->number_of_pixels counts the pixels of all bands
 * together (the last band may be shorter).  With band_pixels = rows * W and band_stride = n_gpus * rows * W these are
 * the interleaved row bands one GPU of n_gpus owns, rendered as one job.  The reference's scheduler deals out
 * contiguous pixel ranges to threads (app/pathtracer.cpp:337-354); every (pixel, sample) has its own random stream
 * here, so a pixel's result does not depend on the job that contains it. */
int ptk_render_bands(PtkScene* scene, PtkFilm* film, const PtkRenderJob* job, int64_t band_pixels, int64_t band_stride);

/* Same, on caller-owned host buffers (the mmap'd temp/raw_*.dat files): uploads the
 * pixel range, renders, downloads, and returns when the buffers are updated. */
int ptk_render_host(PtkScene* scene, const PtkRenderJob* job, double* image, double* position, double* normal,
                    uint64_t* counters);

/* A copy of `scene` on the current device (the one of the last ptk_init on this thread): the tables, triangles and
 * hierarchies are copied device to device, nothing is parsed or built again.  Several GPUs rendering one scene build
 * it once and clone it (the reference builds its one scene once, app/pathtracer.cpp:309-316).  The copy renders
 * the same bits as the original and is destroyed on its own. */
int ptk_scene_clone(PtkScene* scene, PtkScene** out);

/* How the mesh hierarchies of a scene were built (no counterpart in the reference, which prints the whole scene set-up
 * time, app/pathtracer.cpp:309-316): the time of the builds of all meshes together in milliseconds (device time of the
 * build kernels for meshes built on the GPU, wall time of the multi-threaded host build otherwise), how many meshes went
 * which way, and the number of 4-wide nodes.  Any pointer may be NULL. */
int ptk_scene_build_info(PtkScene* scene, double* bvh_build_ms, int32_t* meshes_built_on_gpu, int32_t* meshes_built_on_host,
                         int64_t* bvh_nodes);
/* The same builds by the host's clock, in milliseconds: for a build on the GPU that is its allocations, the upload of
 * the vertices, the waits between the levels of the collapse, the download and the bound on the traversal stack
 * together with the kernels; for a build on the host it is the build itself.  No counterpart in the reference. */
int ptk_scene_build_wall_ms(PtkScene* scene, double* wall_ms);
int ptk_get_stats(PtkScene* scene, PtkStats* stats);
/* Name, as a profiler shows it, of the k_bounce instantiation this scene is traced with (the library compiles one per
 * combination of scene features).  No counterpart in the reference; for benchmark reports. */
const char* ptk_scene_bounce_kernel(PtkScene* scene);
int ptk_reset_stats(PtkScene* scene);
/* No counterpart in the reference (it prints two wall-clock spans, app/pathtracer.cpp:309-372):
 * when enabled, HIP events bracket every kernel family on the library's stream and the
 * *_ms fields of PtkStats accumulate device time.  Also enabled by PTK_PROFILE=1. */
int ptk_set_profiling(int enabled);
/* Tuning knobs, by name, process-wide (no counterpart in the reference).  Results do not depend on any of them.
 * Unknown names and values out of range fail with PTK_ERR_INVALID.
 *   "tail_records": once at most this many paths of a batch are left, one kernel launch follows each of them to
 *     its end instead of running three launches per bounce (0 switches that off; default 1048576, upper limit
 *     16777216).  Scenes with meshes switch when a quarter of the batch's paths are left, but not below this value
 *     nor above four times it; a batch of no more paths than this value switches at this value in every scene.
 *   "launch_ahead": 0 to 7 (default 7), the bounces the host queues beyond the newest record count it has read back;
 *     fewer makes the switch to the tail kernel prompt and leaves the device waiting for the host between bounces.
 *   "batch_lanes": 1 traces the batches of a job one after the other; n = 2 to 4 cuts jobs of 2^21 paths or more into
 *     batches of 1/n of the paths-in-flight limit (2^24 paths at most) and traces n at a time, each on a stream and a
 *     host thread of its own; 0 (default) traces two at a time in scenes with meshes and wherever a job takes more
 *     than one batch anyway, three in scenes with meshes when a job takes more than two.
 *   "gpu_bvh_build": 1 builds the hierarchy of every mesh of 1024 triangles or more on the GPU when the next scenes
 *     are created (a linear BVH over Morton-ordered triangles: built in milliseconds, costlier to walk than the
 *     host's surface-area tree); 0 (default) builds on the host.  A mesh whose build the device has no memory for
 *     (about 370 bytes per triangle) is built on the host all the same.
 *   "gpu_bvh_memory_limit": most device memory, in bytes, one hierarchy build on the GPU may allocate (0, the
 *     default: no limit); a mesh that would need more is built on the host, exactly as when the device runs out.
 *   "traverse_lanes": 1 (default) walks the hierarchies of queued rays with one lane per ray; 4 with four lanes
 *     per ray, one per child box or triangle (slower as measured).
 *   "specialised_kernels": 1 (default) traces a scene with the kernels compiled for what it contains (no
 *     triangles, no texture coordinates, tables small enough for on-chip copies, ...); 0 makes every scene use the
 *     general kernels that scenes with many objects get. */
int ptk_set_option(const char* name, int64_t value);

/* Device self-test (no counterpart in the reference): sends `n_records` (1 to 2^22) path records through the
 * asynchronous memory -> on-chip staging -> register path the shading kernels prefetch their records with, in a
 * kernel with other loads in flight and some lanes switched off, and reports how many arrived different from their
 * source.  0 is the only healthy answer; a regression in the hand-placed wait counts of that path shows here
 * instead of as a rare wrong pixel. */
int ptk_selftest_record_staging(int64_t n_records, int64_t* mismatches);

/* ---- denoise ---------------------------------------------------------------------- */

enum { PTK_DENOISE_ATROUS = 0, PTK_DENOISE_MEDIAN = 1 };

/* DenoisingTask (include/scene.h:31-46). */
typedef struct PtkDenoiseTask {
    int32_t mode;
    int32_t iterations;  /* atrous */
    double sigma_rt, sigma_x, sigma_n;
    int32_t kernel_size; /* median */
    int32_t _pad;
    double threshold;
} PtkDenoiseTask;

/* Replaces denoise() (src/denoise.cpp:203-212): applies the tasks in order to `image`
 * in place, guided by `position` and `normal`; host buffers double[H*W*3].
 * The pipeline is checked before anything runs and rejected with PTK_ERR_INVALID when a median window is even,
 * above 15 or larger than the image (the reference's own check, src/denoise.cpp:149-153), or when the a-trous taps of
 * the last iteration (2^iterations pixels from the centre) reach further than the one reflection at the border covers
 * (2^iterations > min(width, height) - 1): the reference reads outside its buffers in that case. */
int ptk_denoise(const PtkDenoiseTask* tasks, int32_t n_tasks, int32_t width, int32_t height, double* image,
                const double* position, const double* normal);
/* Same on device buffers (image is updated in place; `scratch` is a device buffer of
 * the same size).  Asynchronous on the library's stream. */
int ptk_denoise_device(const PtkDenoiseTask* tasks, int32_t n_tasks, int32_t width, int32_t height,
                       double* d_image, const double* d_position, const double* d_normal, double* d_scratch);
/* ptk_denoise_device without the scratch argument and without any allocation.  No counterpart in the
 * reference: it is for callers that cannot allocate device memory -- code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  The ping-pong buffer is a static array in device memory
 * that every GPU the library runs on holds from load time to process exit: 48 MiB, enough for images of at
 * most 2,097,152 pixels (1920x1080 fits); larger images are refused with PTK_ERR_MEMORY.  The call makes no
 * hipMalloc/hipFree and no allocating library call; same kernels, same results as ptk_denoise_device.
 * All calls on one GPU share that GPU's array: two that overlap in time (different host threads, or streams
 * other than the library's) race.  Asynchronous on the library's stream. */
int ptk_denoise_device_static(const PtkDenoiseTask* tasks, int32_t n_tasks, int32_t width, int32_t height,
                              double* d_image, const double* d_position, const double* d_normal);

#ifdef __cplusplus
}
#endif
#endif /* PTK_H */
