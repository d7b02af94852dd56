/*
 * ptk_host.h -- C ABI of the host-side front end: the scene-directory contract of the
 * reference (`pathtracer <scene_directory>`, `denoiser <scene_directory>`).  Pure host
 * code, no GPU.  It turns settings.json / scene.json / denoising.json, Wavefront OBJ
 * meshes and `.map` textures into the flat PtkSceneDesc that ptk_scene_create() takes,
 * and maps the temp/raw_*.dat accumulation files.
 *
 * Reference interfaces replaced (all in src/scene.cpp unless noted):
 *   load_settings            :83-124     -> ptk_host_load_settings
 *   load_scene               :546-584    -> ptk_host_load_scene
 *   load_denoising_settings  :126-184    -> ptk_host_load_denoising
 *   load_object_model        src/objectunion.cpp:425-443 (OBJ subset + transform)
 *   create_value_map         src/valuemap.cpp:54-91      (.map textures)
 *   create_mmap / close_mmap src/utils.cpp:245-280       -> ptk_host_map_buffer / unmap
 * Error strings follow the reference's messages where it has one.
 */
#ifndef PTK_HOST_H
#define PTK_HOST_H

#include "ptk.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct PtkLoadedScene PtkLoadedScene; /* owns every array its description points to */

const char* ptk_host_last_error(void);

/* Fills `out` with the reference defaults (src/constants.cpp:7-19). */
void ptk_host_default_settings(PtkSettings* out);
/* Defaults overridden by the keys present in the file; validates like load_settings. */
int ptk_host_load_settings(const char* settings_json_path, PtkSettings* out);

/* Parses scene.json.  File names inside it ("models/x.obj", "maps/x.map") are resolved
 * against the current directory first (the reference's behaviour) and then against
 * `asset_root` if given. */
int ptk_host_load_scene(const char* scene_json_path, const PtkSettings* settings, const char* asset_root,
                        PtkLoadedScene** out);
/* Same from a JSON string. */
int ptk_host_load_scene_text(const char* scene_json_text, const PtkSettings* settings, const char* asset_root,
                             PtkLoadedScene** out);
const PtkSceneDesc* ptk_host_scene_desc(const PtkLoadedScene* scene);
/* Name of top-level object `index` as given in scene.json. */
const char* ptk_host_object_name(const PtkLoadedScene* scene, int32_t index);
void ptk_host_free_scene(PtkLoadedScene* scene);

/* Parses denoising.json; "skip" tasks are dropped.  `*tasks` is malloc'd: free() it. */
int ptk_host_load_denoising(const char* denoising_json_path, const PtkSettings* settings, PtkDenoiseTask** tasks,
                            int32_t* n_tasks);

/* Maps `file_size` bytes of `path` read/write and shared, creating the file (mode 0600)
 * and truncating it first when `clear_content` is non-zero.  Returns NULL on failure. */
double* ptk_host_map_buffer(const char* path, size_t file_size, int clear_content, int* fd_out);
void ptk_host_unmap_buffer(double* mapping, size_t file_size, int fd);

/* Writes a `.map` texture: float64 [width, height, dim, data...] (python_utils/get_map.py:34-62). */
int ptk_host_write_map(const char* path, int32_t width, int32_t height, int32_t dim, const double* data);

#ifdef __cplusplus
}
#endif
#endif /* PTK_HOST_H */
