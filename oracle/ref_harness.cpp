// TEST INFRASTRUCTURE -- driver around the UNMODIFIED reference translation units.
//
// This file is compiled together with the reference's own sources, taken from where they
// lie (oracle/Makefile; nothing of the reference is copied into this repository), into
// oracle/_ref/libref.so.  It can only be built where the reference's sources are (the build
// container): the sources do not travel, so what the library produces is committed as
// fixtures under tests/golden/ (the built library itself may be carried along and used as a
// checker or a CPU baseline where it is present; nothing depends on it being there).
//
// What it adds to the reference code:
//   * scene construction from the flat PtkSceneDesc through the reference's public
//     constructors (the reference's own loader, src/scene.cpp, needs nlohmann/json >= 3.6,
//     which its CMake fetches from the network; that file is not on the hot path);
//   * re-seeding of the reference's process-global generators (src/utils.cpp:11-17 are
//     non-static globals) so that single-threaded runs are repeatable;
//   * entry points that call raytrace / compute_pixel_color / raytrace_section /
//     find_closest_hit / denoise and the pure helper functions, and hand back the results.
#include <cstring>
#include <memory>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "bvh.h"
#include "camera.h"
#include "colors.h"
#include "constants.h"
#include "denoise.h"
#include "materials.h"
#include "medium.h"
#include "objects.h"
#include "objectunion.h"
#include "scene.h"
#include "utils.h"
#include "valuemap.h"
#include "vec3.h"

#include "ptk.h"
#include "ptk_host.h"

// Declarations of reference symbols that have external linkage but no header.
extern std::minstd_rand normal_generator;                         // src/utils.cpp:12
extern std::normal_distribution<double> normal_distribution;      // :13
extern std::minstd_rand uniform_generator;                        // :16
extern std::uniform_real_distribution<double> uniform_dist;       // :17

struct PixelData { // app/pathtracer.cpp:20-24
    vec3 pixel_color = vec3(0, 0, 0);
    vec3 pixel_position = vec3(0, 0, 0);
    vec3 pixel_normal = vec3(0, 0, 0);
};
struct Job { // app/pathtracer.cpp:26-30
    size_t start_idx;
    size_t number_of_pixels;
    size_t number_of_samples;
};
struct ThreadContext { // app/pathtracer.cpp:32-37
    size_t* thread_progress;
    size_t thread_idx;
    std::mutex* pixel_section_mutexes;
    size_t* pixel_counters;
};
PixelData compute_pixel_color(const size_t x, const size_t y, const Scene& scene, const size_t number_of_samples);
void raytrace_section(const Job& job, const Scene& scene, PixelBuffers& buffers, ThreadContext& thread_context);
void thread_job(std::vector<Job>* job_queue, const Scene& scene, PixelBuffers buffers, ThreadContext thread_context);

namespace {
thread_local std::string g_error;

int set_error(int code, const std::string& m) {
    g_error = m;
    return code;
}

vec3 to_vec3(const double* p) {
    return vec3(p[0], p[1], p[2]);
}

void apply_settings(const PtkSettings& s) {
    constants::WIDTH = static_cast<size_t>(s.width);
    constants::HEIGHT = static_cast<size_t>(s.height);
    constants::samples_per_pixel = static_cast<size_t>(s.samples_per_pixel);
    constants::samples_per_iteration = static_cast<size_t>(s.samples_per_iteration);
    constants::max_recursion_depth = s.max_recursion_depth;
    constants::min_recursion_steps = s.min_recursion_steps;
    constants::number_of_threads = static_cast<size_t>(s.number_of_threads);
    constants::use_gamma_correction = s.use_gamma_correction != 0;
    constants::bvh_leaf_size = s.bvh_leaf_size;
    constants::bvh_n_axis_splits = s.bvh_n_axis_splits;
    constants::enable_next_event_estimation = s.enable_next_event_estimation != 0;
    constants::enable_anti_aliasing = s.enable_anti_aliasing != 0;
}
} // namespace

struct RefScene {
    PtkSettings settings{};
    Scene scene{};
    std::vector<Object const*> objects;
    std::vector<std::vector<Triangle*>> mesh_faces; // per object: triangles in OBJ face order
    std::vector<std::vector<int64_t>> face_of_primitive;
};

namespace {

Scene build_scene(const PtkSceneDesc& d, RefScene* keep) {
    apply_settings(d.settings);
    PointerManager* manager = new PointerManager();
    std::vector<ValueMap const*> maps;
    for (int i = 0; i < d.n_maps; i++) {
        const PtkMap& m = d.maps[i];
        size_t n = static_cast<size_t>(m.width) * static_cast<size_t>(m.height) * static_cast<size_t>(m.dim);
        double* data = new double[n];
        std::memcpy(data, d.map_data + m.offset, n * sizeof(double));
        ValueMap const* map;
        if (m.dim == 3) map = new ValueMap3D(data, m.width, m.height);
        else map = new ValueMap1D(data, m.width, m.height);
        maps.push_back(map);
        manager->add_valuemap(map);
    }
    std::vector<Medium const*> media;
    for (int i = 0; i < d.n_media; i++) {
        const PtkMedium& m = d.media[i];
        Medium const* medium;
        if (m.type == PTK_MEDIUM_BEERS_LAW)
            medium = new BeersLawMedium(to_vec3(m.scattering), to_vec3(m.absorption), to_vec3(m.emission), m.refractive_index);
        else
            medium = new HomogenousScatteringMedium(to_vec3(m.scattering), to_vec3(m.absorption), to_vec3(m.emission),
                                                    m.refractive_index);
        media.push_back(medium);
        manager->add_medium(medium);
    }
    std::vector<Material const*> materials;
    for (int i = 0; i < d.n_materials; i++) {
        const PtkMaterial& m = d.materials[i];
        MaterialData data;
        data.is_light_source = m.is_light_source != 0;
        data.is_dielectric = m.is_dielectric != 0;
        data.surface_refractive_index = m.surface_refractive_index;
        data.extinction_coefficient = m.extinction_coefficient;
        if (m.albedo_map >= 0) data.albedo_map = static_cast<ValueMap3D const*>(maps[static_cast<size_t>(m.albedo_map)]);
        if (m.emission_color_map >= 0)
            data.emission_color_map = static_cast<ValueMap3D const*>(maps[static_cast<size_t>(m.emission_color_map)]);
        if (m.light_intensity_map >= 0)
            data.light_intensity_map = static_cast<ValueMap1D const*>(maps[static_cast<size_t>(m.light_intensity_map)]);
        if (m.roughness_map >= 0)
            data.roughness_map = static_cast<ValueMap1D const*>(maps[static_cast<size_t>(m.roughness_map)]);
        if (m.transparency_map >= 0)
            data.transparency_map = static_cast<ValueMap1D const*>(maps[static_cast<size_t>(m.transparency_map)]);
        if (m.internal_medium >= 0) data.internal_medium = media[static_cast<size_t>(m.internal_medium)];
        if (m.external_medium >= 0) data.external_medium = media[static_cast<size_t>(m.external_medium)];
        Material const* material = nullptr;
        switch (m.type) {
            case PTK_MAT_DIFFUSE: material = new DiffuseMaterial(data); break;
            case PTK_MAT_REFLECTIVE: material = new ReflectiveMaterial(data); break;
            case PTK_MAT_TRANSPARENT: material = new TransparentMaterial(data); break;
            case PTK_MAT_GLOSSY: material = new GlossyMaterial(data); break;
            case PTK_MAT_METALLIC_MICROFACET: material = new MetallicMicrofacetMaterial(data); break;
            case PTK_MAT_REFLECTIVE_MICROFACET: material = new ReflectiveMicrofacetMaterial(data); break;
            case PTK_MAT_TRANSPARENT_MICROFACET: material = new TransparentMicrofacetMaterial(data); break;
            default: throw std::runtime_error("unknown material type");
        }
        materials.push_back(material);
        manager->add_material(material);
    }
    Object const** objects = new Object const*[static_cast<size_t>(d.n_objects)];
    for (int i = 0; i < d.n_objects; i++) {
        const PtkObject& o = d.objects[i];
        Material const* material = materials[static_cast<size_t>(o.material)];
        std::vector<Triangle*> faces;
        switch (o.type) {
            case PTK_OBJ_SPHERE: objects[i] = new Sphere(to_vec3(o.position), o.radius, material); break;
            case PTK_OBJ_PLANE: objects[i] = new Plane(to_vec3(o.position), to_vec3(o.v1), to_vec3(o.v2), material); break;
            case PTK_OBJ_RECTANGLE:
                objects[i] = new Rectangle(to_vec3(o.position), to_vec3(o.v1), to_vec3(o.v2), o.L1, o.L2, material);
                break;
            case PTK_OBJ_TRIANGLE:
                objects[i] = new Triangle(to_vec3(o.position), to_vec3(o.v1), to_vec3(o.v2), material);
                break;
            case PTK_OBJ_MESH: {
                Object** triangles = new Object*[static_cast<size_t>(o.n_triangles)];
                for (int64_t t = 0; t < o.n_triangles; t++) {
                    int64_t g = o.first_triangle + t;
                    const double* p = d.tri_vertices + 9 * g;
                    Triangle* tri = new Triangle(to_vec3(p), to_vec3(p + 3), to_vec3(p + 6), material);
                    if (d.tri_flags[g] & PTK_TRI_HAS_UV) {
                        const double* uv = d.tri_uv + 6 * g;
                        tri->set_vertex_UV(vec3(uv[0], uv[1], 0), vec3(uv[2], uv[3], 0), vec3(uv[4], uv[5], 0));
                    }
                    if (d.tri_flags[g] & PTK_TRI_SMOOTH) {
                        const double* n = d.tri_normals + 9 * g;
                        tri->set_vertex_normals(to_vec3(n), to_vec3(n + 3), to_vec3(n + 6));
                    }
                    triangles[t] = tri;
                    faces.push_back(tri);
                }
                objects[i] = new ObjectUnion(triangles, static_cast<int>(o.n_triangles), true);
                break;
            }
            default: throw std::runtime_error("unknown object type");
        }
        if (keep) {
            keep->mesh_faces.push_back(faces);
            std::vector<int64_t> inverse(faces.size());
            for (size_t f = 0; f < faces.size(); f++) inverse[static_cast<size_t>(faces[f]->primitive_ID)] = static_cast<int64_t>(f);
            keep->face_of_primitive.push_back(inverse);
        }
    }
    const PtkCamera& c = d.camera;
    std::string mode = c.dof_mode == PTK_DOF_CIRCLE ? "circle" : (c.dof_mode == PTK_DOF_SQUARE ? "square" : "none");
    Scene scene;
    scene.objects = objects;
    scene.number_of_objects = d.n_objects;
    scene.camera = Camera(to_vec3(c.position), c.orientation_x, c.orientation_y, c.orientation_z, c.camera_width,
                          c.focal_length, mode, c.aperture_size, c.focus_distance);
    scene.pointer_manager = manager;
    scene.medium = media[static_cast<size_t>(d.background_medium)];
    scene.background_color = to_vec3(d.background_color);
    return scene;
}

PtkSettings g_loaded_settings;
bool g_have_settings = false;

} // namespace

// PointerManager (include/scene.h:13-29) is declared in a reference header but defined in the
// loader file that cannot be compiled here; it only owns pointers, so it is defined here.
PointerManager::~PointerManager() {
    for (ValueMap const* p : valuemaps) delete p;
    for (Medium const* p : media) delete p;
    for (Material const* p : materials) delete p;
}
void PointerManager::add_valuemap(ValueMap const* const map) {
    valuemaps.push_back(map);
}
void PointerManager::add_medium(Medium const* const medium) {
    media.push_back(medium);
}
void PointerManager::add_material(Material const* const material) {
    materials.push_back(material);
}

// The two loader entry points the reference's main() calls (include/scene.h:65,67), here
// backed by this repository's host loader so that the reference executable links.
void load_settings(const std::string& file_path) {
    if (ptk_host_load_settings(file_path.c_str(), &g_loaded_settings) != PTK_OK)
        throw std::runtime_error(ptk_host_last_error());
    g_have_settings = true;
    apply_settings(g_loaded_settings);
}

Scene load_scene(const std::string& file_path) {
    if (!g_have_settings) ptk_host_default_settings(&g_loaded_settings);
    PtkLoadedScene* loaded = nullptr;
    if (ptk_host_load_scene(file_path.c_str(), &g_loaded_settings, nullptr, &loaded) != PTK_OK)
        throw std::runtime_error(ptk_host_last_error());
    Scene scene = build_scene(*ptk_host_scene_desc(loaded), nullptr);
    ptk_host_free_scene(loaded);
    return scene;
}

extern "C" {

const char* ref_last_error(void) {
    return g_error.c_str();
}

int ref_scene_create(const PtkSceneDesc* desc, RefScene** out) {
    if (!desc || !out) return set_error(PTK_ERR_INVALID, "ref_scene_create: null argument");
    try {
        std::unique_ptr<RefScene> s(new RefScene());
        s->settings = desc->settings;
        s->scene = build_scene(*desc, s.get());
        *out = s.release();
        return PTK_OK;
    }
    catch (const std::exception& e) {
        return set_error(PTK_ERR_INVALID, e.what());
    }
}

void ref_scene_destroy(RefScene* s) {
    if (!s) return;
    for (int i = 0; i < s->scene.number_of_objects; i++) delete s->scene.objects[i];
    delete[] s->scene.objects;
    delete s->scene.pointer_manager;
    delete s;
}

// Re-seeds the reference's three generators; afterwards a single-threaded run is repeatable.
void ref_seed(uint32_t uniform_seed, uint32_t normal_seed, uint32_t rand_seed) {
    uniform_generator.seed(uniform_seed);
    normal_generator.seed(normal_seed);
    uniform_dist.reset();
    normal_distribution.reset();
    srand(rand_seed);
}

int ref_trace_primary(RefScene* s, const PtkRenderJob* job, int32_t* object_index, int32_t* primitive_index,
                      double* distance, double* position, double* normal) {
    if (!s || !job) return set_error(PTK_ERR_INVALID, "ref_trace_primary: null argument");
    apply_settings(s->settings);
    for (int64_t i = 0; i < job->number_of_pixels; i++) {
        size_t idx = static_cast<size_t>(job->start_idx + i);
        size_t x = idx % constants::WIDTH;
        size_t y = constants::HEIGHT - idx / constants::WIDTH;
        double new_x = static_cast<double>(x), new_y = static_cast<double>(y);
        if (constants::enable_anti_aliasing) { // app/pathtracer.cpp:185-188
            new_x += random_normal() / 3.0;
            new_y += random_normal() / 3.0;
        }
        Ray ray = s->scene.camera.make_ray(new_x, new_y);
        Hit hit;
        bool found = find_closest_hit(hit, ray, s->scene.objects, s->scene.number_of_objects);
        int64_t face = 0;
        if (found && !s->face_of_primitive[static_cast<size_t>(hit.intersected_object_index)].empty())
            face = s->face_of_primitive[static_cast<size_t>(hit.intersected_object_index)][static_cast<size_t>(hit.primitive_ID)];
        if (object_index) object_index[i] = found ? hit.intersected_object_index : -1;
        if (primitive_index) primitive_index[i] = found ? static_cast<int32_t>(face) : 0;
        if (distance) distance[i] = found ? hit.distance : constants::max_ray_distance;
        for (int j = 0; j < 3; j++) {
            if (position) position[3 * i + j] = found ? hit.intersection_point[j] : 0.0;
            if (normal) normal[3 * i + j] = found ? hit.normal_vector[j] : 0.0;
        }
    }
    return PTK_OK;
}

int ref_trace_samples(RefScene* s, const PtkRenderJob* job, double* color, double* position, double* normal) {
    if (!s || !job) return set_error(PTK_ERR_INVALID, "ref_trace_samples: null argument");
    apply_settings(s->settings);
    for (int64_t i = 0; i < job->number_of_pixels; i++) {
        size_t idx = static_cast<size_t>(job->start_idx + i);
        size_t x = idx % constants::WIDTH;
        size_t y = constants::HEIGHT - idx / constants::WIDTH;
        for (int32_t k = 0; k < job->number_of_samples; k++) {
            PixelData d = compute_pixel_color(x, y, s->scene, 1); // one sample: jitter, make_ray, raytrace, /1.0
            int64_t o = 3 * (i * job->number_of_samples + k);
            for (int j = 0; j < 3; j++) {
                if (color) color[o + j] = d.pixel_color[j];
                if (position) position[o + j] = d.pixel_position[j];
                if (normal) normal[o + j] = d.pixel_normal[j];
            }
        }
    }
    return PTK_OK;
}

// The reference's scheduler (app/pathtracer.cpp:334-365) for one image: `iterations` rounds of
// `threads` pixel-range jobs of `samples_per_iteration` samples, run by its own thread_job.
int ref_render(RefScene* s, int32_t threads, double* image, double* position, double* normal) {
    if (!s || !image || !position || !normal || threads < 1) return set_error(PTK_ERR_INVALID, "ref_render: bad argument");
    apply_settings(s->settings);
    constants::number_of_threads = static_cast<size_t>(threads);
    size_t pixel_count = constants::WIDTH * constants::HEIGHT;
    PixelBuffers buffers{image, position, normal};
    std::memset(image, 0, pixel_count * 3 * sizeof(double));
    std::memset(position, 0, pixel_count * 3 * sizeof(double));
    std::memset(normal, 0, pixel_count * 3 * sizeof(double));
    size_t pixels_per_thread = ceil_division(pixel_count, constants::number_of_threads);
    std::vector<size_t> thread_progress(static_cast<size_t>(threads), 0);
    std::vector<Job> job_queue;
    size_t iterations = ceil_division(constants::samples_per_pixel, constants::samples_per_iteration);
    for (size_t j = 0; j < iterations; j++) {
        for (size_t i = 0; i < constants::number_of_threads; i++) {
            size_t start_idx = pixels_per_thread * i;
            if (start_idx > pixel_count) break;
            Job job;
            job.start_idx = start_idx;
            job.number_of_pixels = std::min<size_t>(pixels_per_thread, pixel_count - start_idx);
            job.number_of_samples = std::min<size_t>(constants::samples_per_iteration,
                                                     constants::samples_per_pixel - j * constants::samples_per_iteration);
            job_queue.push_back(job);
        }
    }
    std::vector<std::mutex> mutexes(static_cast<size_t>(threads));
    std::vector<size_t> counters(pixel_count, 0);
    std::vector<std::thread> pool;
    for (size_t i = 0; i < static_cast<size_t>(threads); i++) {
        ThreadContext ctx{thread_progress.data(), i, mutexes.data(), counters.data()};
        pool.emplace_back(thread_job, &job_queue, std::cref(s->scene), buffers, ctx);
    }
    for (auto& t : pool) t.join();
    return PTK_OK;
}

int ref_denoise(const PtkDenoiseTask* tasks, int32_t n_tasks, int32_t W, int32_t H, double* image,
                double* position, double* normal) {
    constants::WIDTH = static_cast<size_t>(W);
    constants::HEIGHT = static_cast<size_t>(H);
    std::vector<DenoisingTask> pipeline;
    for (int32_t i = 0; i < n_tasks; i++) {
        DenoisingTask t;
        if (tasks[i].mode == PTK_DENOISE_ATROUS) {
            t.mode = "atrous";
            t.params = AtrousParamters{tasks[i].iterations, tasks[i].sigma_rt, tasks[i].sigma_x, tasks[i].sigma_n};
        }
        else {
            t.mode = "median";
            t.params = MedianFilterParamters{tasks[i].kernel_size, tasks[i].threshold};
        }
        pipeline.push_back(t);
    }
    PixelBuffers buffers{image, position, normal};
    denoise(buffers, pipeline);
    return PTK_OK;
}

// ---- pure helper functions of the reference, for known-answer tables -------------------
double ref_fresnel_dielectric(double c, double n1, double n2) { return fresnel_dielectric(c, n1, n2); }
double ref_fresnel_conductor(double c, double n1, double k1, double n2, double k2) { return fresnel_conductor(c, n1, k1, n2, k2); }
double ref_schlick_fresnel(double c, double n1, double n2) { return schlick_fresnel(c, n1, 0, n2, 0); }
double ref_beckmann_D(const double h[3], const double n[3], double alpha) {
    GlossyMaterial m{MaterialData()};
    return m.D(to_vec3(h), to_vec3(n), alpha);
}
double ref_smith_G1(const double h[3], const double n[3], const double v[3], double alpha) {
    GlossyMaterial m{MaterialData()};
    return m.G1(to_vec3(h), to_vec3(n), to_vec3(v), alpha);
}
void ref_rotate(const double p[3], double a, double b, double g, double out[3]) {
    vec3 r = rotate(to_vec3(p), a, b, g);
    for (int i = 0; i < 3; i++) out[i] = r[i];
}
void ref_refract(const double i[3], const double n[3], double eta, double out[3]) {
    vec3 r = refract_vector(to_vec3(i), to_vec3(n), eta);
    for (int k = 0; k < 3; k++) out[k] = r[k];
}
int ref_solve_quadratic(double b, double c, double* distance) {
    double d = 0;
    bool ok = solve_quadratic(b, c, d);
    if (ok && distance) *distance = d;
    return ok ? 1 : 0;
}

} // extern "C"
