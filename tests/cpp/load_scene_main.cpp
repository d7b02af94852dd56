// Loads one scene directory with the host loader and reports what happened: "loaded <objects> <triangles>" or
// "error: <message>", exit status 0 either way.  Built with the address and undefined-behaviour sanitizers
// (tools/sanitize_host.sh), it is what tools/fuzz_loader.py feeds its mutated directories to: a crash or a sanitizer
// report ends the process with another status.  usage: load_scene_main <scene_directory>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "ptk_host.h"

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s <scene_directory>\n", argv[0]);
        return 2;
    }
    const std::string dir = argv[1];
    PtkSettings settings;
    ptk_host_default_settings(&settings);
    if (ptk_host_load_settings((dir + "/settings.json").c_str(), &settings) != 0) {
        std::printf("error: %s\n", ptk_host_last_error());
        return 0;
    }
    PtkLoadedScene* scene = nullptr;
    if (ptk_host_load_scene((dir + "/scene.json").c_str(), &settings, dir.c_str(), &scene) != 0) {
        std::printf("error: %s\n", ptk_host_last_error());
        return 0;
    }
    const PtkSceneDesc* d = ptk_host_scene_desc(scene);
    // every array the description points to is read once, so that a pointer past its allocation shows
    double sum = 0;
    for (int64_t i = 0; i < d->n_triangles * 9; i++) sum += d->tri_vertices[i];
    for (int64_t i = 0; i < d->n_map_values; i++) sum += d->map_data[i];
    for (int32_t i = 0; i < d->n_objects; i++) sum += d->objects[i].type + (ptk_host_object_name(scene, i) ? 1 : 0);
    for (int32_t i = 0; i < d->n_materials; i++) sum += d->materials[i].type;
    for (int32_t i = 0; i < d->n_media; i++) sum += d->media[i].type;
    PtkDenoiseTask* tasks = nullptr;
    int32_t n_tasks = 0;
    if (ptk_host_load_denoising((dir + "/denoising.json").c_str(), &settings, &tasks, &n_tasks) == 0 && n_tasks > 0) std::free(tasks);
    std::printf("loaded %d %lld %g\n", d->n_objects, static_cast<long long>(d->n_triangles), sum != sum ? 0.0 : 1.0);
    ptk_host_free_scene(scene);
    return 0;
}
