// Host front end: scene-directory files -> flat PtkSceneDesc.  See include/ptk_host.h.
//
// Behaviour follows the reference loaders so that an unchanged scene directory loads to
// the same scene: src/scene.cpp (JSON schema, defaults, error messages),
// src/objectunion.cpp:144-443 (OBJ subset and its whitespace rules, centre/scale/rotate
// normalisation), src/valuemap.cpp:54-91 (.map files), src/utils.cpp:245-280 (mmap).
#include "ptk_host.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "json_mini.h"

using ptkjson::Value;

namespace {

thread_local std::string g_error;

int fail(int code, const std::string& message) {
    g_error = message;
    return code;
}

struct LoadError : std::runtime_error {
    int code;
    LoadError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

bool read_text_file(const std::string& path, std::string& out) {
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) return false;
    std::ostringstream ss;
    ss << f.rdbuf();
    out = ss.str();
    return true;
}

bool file_exists(const std::string& path) {
    struct stat st;
    return stat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

// -------------------------------------------------------------------------------------
// small double[3] helpers; operation order matches the reference's vec3 so that derived
// vertex data is bit-identical to what the reference would hold
// -------------------------------------------------------------------------------------
struct D3 {
    double x, y, z;
};

// YZX Tait-Bryan rotation in degrees (src/utils.cpp:143-173).
D3 rotate_yzx(const D3& p, double alpha, double beta, double gamma) {
    double a = alpha * M_PI / 180, b = beta * M_PI / 180, g = gamma * M_PI / 180;
    double ca = std::cos(a), sa = std::sin(a);
    double cb = std::cos(b), sb = std::sin(b);
    double cg = std::cos(g), sg = std::sin(g);
    D3 r;
    r.x = p.x * (ca * cb) + p.y * (sa * sg - ca * cg * sb) + p.z * (cg * sa + ca * sb * sg);
    r.y = p.x * (sb) + p.y * (cb * cg) + p.z * (-cb * sg);
    r.z = p.x * (-cb * sa) + p.y * (ca * sg + cg * sa * sb) + p.z * (ca * cg - sa * sb * sg);
    return r;
}

// sRGB -> linear (include/colors.h:21-24).
double srgb_to_linear(double x) {
    return x <= 0.04045 ? x / 12.92 : std::pow((x + 0.055) / 1.055, 2.4);
}

} // namespace

struct PtkLoadedScene {
    PtkSceneDesc desc{};
    std::vector<PtkObject> objects;
    std::vector<std::string> object_names;
    std::vector<PtkMaterial> materials;
    std::vector<PtkMedium> media;
    std::vector<PtkMap> maps;
    std::vector<double> map_data;
    std::vector<double> tri_vertices, tri_uv, tri_normals;
    std::vector<uint8_t> tri_flags;
};

namespace {

// -------------------------------------------------------------------------------------
// settings.json
// -------------------------------------------------------------------------------------
template<typename T>
void load_one_setting(const Value& data, const char* key, T& setting) {
    if (data.contains(key)) {
        const Value& v = data.at(key);
        setting = static_cast<T>(v.kind == Value::Bool ? (v.b ? 1 : 0) : v.as_number());
    }
}

void default_settings(PtkSettings& s) {
    s.width = 1000;
    s.height = 1000;
    s.samples_per_pixel = 10;
    s.samples_per_iteration = 16;
    s.max_recursion_depth = 100;
    s.min_recursion_steps = 3;
    s.number_of_threads = 1;
    s.use_gamma_correction = 1;
    s.bvh_leaf_size = 4;
    s.bvh_n_axis_splits = 16;
    s.enable_next_event_estimation = 1;
    s.enable_anti_aliasing = 1;
}

void parse_settings(const std::string& text, PtkSettings& s) {
    Value data = ptkjson::parse(text);
    default_settings(s);
    load_one_setting(data, "WIDTH", s.width);
    load_one_setting(data, "HEIGHT", s.height);
    load_one_setting(data, "samples_per_pixel", s.samples_per_pixel);
    load_one_setting(data, "samples_per_iteration", s.samples_per_iteration);
    if (s.samples_per_iteration <= 0)
        throw LoadError(PTK_ERR_FORMAT, "Setting 'samples_per_iteration' must be a positive number!");
    load_one_setting(data, "max_recursion_depth", s.max_recursion_depth);
    load_one_setting(data, "min_recursion_steps", s.min_recursion_steps);
    load_one_setting(data, "number_of_threads", s.number_of_threads);
    if (s.number_of_threads <= 0) throw LoadError(PTK_ERR_FORMAT, "'number_of_threads' must be 1 or greater.");
    int max_threads = std::max<int>(static_cast<int>(std::thread::hardware_concurrency()) - 1, 1);
    if (s.number_of_threads > max_threads) {
        std::clog << "Warning: The requested number of threads was greater than the maximum concurrency available. "
                     "Capping at "
                  << max_threads << ".\n";
        s.number_of_threads = max_threads;
    }
    load_one_setting(data, "use_gamma_correction", s.use_gamma_correction);
    load_one_setting(data, "bvh_leaf_size", s.bvh_leaf_size);
    if (s.bvh_leaf_size < 1) throw LoadError(PTK_ERR_FORMAT, "'bvh_leaf_size' must be 1 or greater.");
    load_one_setting(data, "bvh_n_axis_splits", s.bvh_n_axis_splits);
    if (s.bvh_n_axis_splits < 1) throw LoadError(PTK_ERR_FORMAT, "'bvh_n_axis_splits' must be 1 or greater.");
    load_one_setting(data, "enable_next_event_estimation", s.enable_next_event_estimation);
    load_one_setting(data, "enable_anti_aliasing", s.enable_anti_aliasing);
    if (s.width <= 0 || s.height <= 0) throw LoadError(PTK_ERR_FORMAT, "'WIDTH' and 'HEIGHT' must be positive.");
}

// -------------------------------------------------------------------------------------
// scene.json helpers
// -------------------------------------------------------------------------------------
void require_field(const Value& data, const std::string& key) {
    if (!data.contains(key)) throw LoadError(PTK_ERR_FORMAT, "JSON structure missing key '" + key + "'");
}

void get_vec3_param(const Value& data, const std::string& key, double out[3]) {
    const Value& v = data.at(key);
    if (v.kind != Value::Array || v.size() != 3)
        throw LoadError(PTK_ERR_FORMAT, "Data array'" + v.dump() + "' should contain 3 elements, but contains " +
                                            std::to_string(v.size()));
    for (int i = 0; i < 3; i++) out[i] = v.items[static_cast<size_t>(i)].as_number();
}

// `.map` file: float64 [width, height, dim, data...] (src/valuemap.cpp:54-91).
void load_map_file(const std::string& file_name, const std::string& asset_root, bool gamma_correct, int expect_dim,
                   PtkLoadedScene& scene, PtkMap& map) {
    std::string path = file_name;
    if (!file_exists(path) && !asset_root.empty()) path = asset_root + "/" + file_name;
    std::ifstream file(path, std::ios::binary);
    if (!file) throw LoadError(PTK_ERR_IO, "Could not open file '" + file_name + "'");
    file.seekg(0, std::ios::end);
    std::streamsize size = file.tellg();
    file.seekg(0, std::ios::beg);
    size_t n = static_cast<size_t>(size) / sizeof(double);
    std::vector<double> data(n);
    if (n < 3 || !file.read(reinterpret_cast<char*>(data.data()), static_cast<std::streamsize>(n * sizeof(double))))
        throw LoadError(PTK_ERR_IO, "Error reading file '" + file_name + "'");
    const std::string bad = "File '" + file_name + "' does not follow the expected format and cannot be loaded";
    if (data[0] < 0 || data[1] < 0 || data[2] < 0) throw LoadError(PTK_ERR_FORMAT, bad);
    int width = static_cast<int>(data[0]);
    int height = static_cast<int>(data[1]);
    size_t dim = static_cast<size_t>(data[2]);
    size_t num_values = static_cast<size_t>(width) * static_cast<size_t>(height) * dim;
    if (num_values != n - 3 || width < 1 || height < 1) throw LoadError(PTK_ERR_FORMAT, bad);
    if (static_cast<int>(dim) < expect_dim) // the reference would read past the end of such a file's data
        throw LoadError(PTK_ERR_FORMAT, bad);
    map.width = width;
    map.height = height;
    // a file with more channels than the map type reads is indexed with the map type's stride by
    // the reference (src/valuemap.cpp:40,50); keep the data as stored and the type's stride
    map.dim = expect_dim;
    map.offset = static_cast<int64_t>(scene.map_data.size());
    for (size_t i = 3; i < n; i++) scene.map_data.push_back(gamma_correct ? srgb_to_linear(data[i]) : data[i]);
}

// -------------------------------------------------------------------------------------
// Wavefront OBJ subset (src/objectunion.cpp:144-443)
// -------------------------------------------------------------------------------------
int count_char(const std::string& line, char c) {
    int n = 0;
    for (char ch : line)
        if (ch == c) n++;
    return n;
}

// n-th field when splitting on every single occurrence of `delimiter`
// (get_nth_word, src/objectunion.cpp:154-168): consecutive delimiters give empty fields.
std::string nth_field(const std::string& line, char delimiter, int n) {
    if (n > count_char(line, delimiter)) return "";
    size_t start = 0;
    for (int i = 0; i < n; i++) {
        size_t pos = line.find(delimiter, start);
        start = pos + 1;
    }
    size_t end = line.find(delimiter, start);
    if (end == std::string::npos) return line.substr(start);
    return line.substr(start, end - start);
}

double parse_double(const std::string& word, const std::string& file_name) {
    const char* begin = word.c_str();
    char* end = nullptr;
    double d = std::strtod(begin, &end);
    if (end == begin) throw LoadError(PTK_ERR_FORMAT, "Invalid number '" + word + "' in file '" + file_name + "'");
    return d;
}

struct ObjTransform {
    bool move_object = false, rotate_object = false, scale_object = false;
    double center[3] = {0, 0, 0};
    double orientation[3] = {0, 0, 0}; // (Y, Z, X) as get_rotation_parameters returns it
    double size = 0;
};

struct CornerRef {
    bool v_ok = false, uv_ok = false, n_ok = false;
    long v = 0, uv = 0, n = 0;
};

CornerRef parse_corner(const std::string& word) {
    CornerRef c;
    std::string v = nth_field(word, '/', 0), uv = nth_field(word, '/', 1), n = nth_field(word, '/', 2);
    if (!v.empty()) {
        c.v = std::strtol(v.c_str(), nullptr, 10) - 1;
        c.v_ok = true;
    }
    if (!uv.empty()) {
        c.uv = std::strtol(uv.c_str(), nullptr, 10) - 1;
        c.uv_ok = true;
    }
    if (!n.empty()) {
        c.n = std::strtol(n.c_str(), nullptr, 10) - 1;
        c.n_ok = true;
    }
    return c;
}

// Runs work(part, n_parts) for part = 0 .. n_parts - 1 on threads of their own; a LoadError any of them throws is
// kept, and the one of the lowest part (the first in file order) is thrown again here once all have ended.
template<typename Work>
void run_parts(int n_parts, Work work) {
    std::vector<std::unique_ptr<LoadError>> errors(static_cast<size_t>(n_parts));
    auto guarded = [&](int part) {
        try {
            work(part, n_parts);
        }
        catch (const LoadError& e) {
            errors[static_cast<size_t>(part)].reset(new LoadError(e));
        }
    };
    std::vector<std::thread> threads;
    for (int part = 1; part < n_parts; part++) threads.emplace_back(guarded, part);
    guarded(0);
    for (std::thread& t : threads) t.join();
    for (const auto& e : errors)
        if (e) throw LoadError(*e);
}

// A file mapped read-only (empty files map to nothing).
struct MappedFile {
    const char* data = nullptr;
    size_t size = 0;
    int fd = -1;
    explicit MappedFile(const std::string& path) {
        fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) return;
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
            close(fd);
            fd = -1;
            return;
        }
        size = static_cast<size_t>(st.st_size);
        if (size > 0) {
            void* m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) {
                close(fd);
                fd = -1;
                size = 0;
                return;
            }
            data = static_cast<const char*>(m);
        }
    }
    ~MappedFile() {
        if (data) munmap(const_cast<char*>(data), size);
        if (fd >= 0) close(fd);
    }
    MappedFile(const MappedFile&) = delete;
    MappedFile& operator=(const MappedFile&) = delete;
    bool is_open() const { return fd >= 0; }
};

// The reference reads the file line by line, three times (src/objectunion.cpp:170-253,378-423).  The format is
// line-oriented and every line stands for itself, so the file is cut at line ends into one part per thread: each part
// collects its `v`, `vt`, `vn` and `f` lines in order, and the parts are joined in file order, which gives exactly
// what one pass over the whole file gives (a 1 GB file of 10,485,760 triangles: 8.5 s on one thread, which was two
// thirds of the whole scene set-up).  Lines are what std::getline makes of the file: cut at '\n', a '\r' before it
// stays part of the line, a last line without '\n' counts.
void load_obj(const std::string& file_name, const std::string& asset_root, bool smooth, const ObjTransform& tf,
              PtkLoadedScene& scene, int64_t& first_triangle, int64_t& n_triangles) {
    std::string path = file_name;
    if (!file_exists(path) && !asset_root.empty()) path = asset_root + "/" + file_name;
    MappedFile file(path);
    if (!file.is_open()) throw LoadError(PTK_ERR_IO, "Could not open file '" + file_name + "'");

    // one part per hardware thread, at most 16, for files of a megabyte or more (PTK_OBJ_PARSE_PARTS in the environment
    // sets the number of parts, 1 to 64, whatever the size of the file: what the tests use to take small files apart)
    int n_parts = file.size < (size_t(1) << 20) ? 1 : static_cast<int>(std::max(1u, std::min(16u, std::thread::hardware_concurrency())));
    if (const char* forced = std::getenv("PTK_OBJ_PARSE_PARTS")) {
        const int wanted = std::atoi(forced);
        if (wanted >= 1 && wanted <= 64) n_parts = wanted;
    }
    // part k covers [cut[k], cut[k + 1]): cuts lie just behind a '\n'
    std::vector<size_t> cut(static_cast<size_t>(n_parts) + 1, file.size);
    cut[0] = 0;
    for (int k = 1; k < n_parts; k++) {
        size_t at = std::max(cut[static_cast<size_t>(k) - 1], file.size / static_cast<size_t>(n_parts) * static_cast<size_t>(k));
        while (at < file.size && file.data[at] != '\n') at++;
        cut[static_cast<size_t>(k)] = std::min(at + 1, file.size);
    }
    struct Part {
        std::vector<D3> vertices, uvs, normals;
        std::vector<std::string> faces;
    };
    std::vector<Part> parts(static_cast<size_t>(n_parts));
    run_parts(n_parts, [&](int k, int) {
        Part& part = parts[static_cast<size_t>(k)];
        std::string line;
        for (size_t at = cut[static_cast<size_t>(k)], end = cut[static_cast<size_t>(k) + 1]; at < end;) {
            const void* newline = std::memchr(file.data + at, '\n', end - at);
            const size_t line_end = newline ? static_cast<size_t>(static_cast<const char*>(newline) - file.data) : end;
            line.assign(file.data + at, line_end - at);
            at = line_end + 1;
            std::string first = nth_field(line, ' ', 0);
            if (first == "v") {
                part.vertices.push_back({parse_double(nth_field(line, ' ', 1), file_name),
                                         parse_double(nth_field(line, ' ', 2), file_name),
                                         parse_double(nth_field(line, ' ', 3), file_name)});
            }
            else if (first == "vt") {
                part.uvs.push_back({parse_double(nth_field(line, ' ', 1), file_name),
                                    parse_double(nth_field(line, ' ', 2), file_name), 0.0});
            }
            else if (first == "vn") {
                part.normals.push_back({parse_double(nth_field(line, ' ', 1), file_name),
                                        parse_double(nth_field(line, ' ', 2), file_name),
                                        parse_double(nth_field(line, ' ', 3), file_name)});
            }
            else if (first == "f") {
                part.faces.push_back(line);
            }
        }
    });
    std::vector<D3> vertices, uvs, normals;
    std::vector<std::string> faces;
    if (n_parts == 1) {
        vertices.swap(parts[0].vertices);
        uvs.swap(parts[0].uvs);
        normals.swap(parts[0].normals);
        faces.swap(parts[0].faces);
    }
    else {
        for (Part& part : parts) {
            vertices.insert(vertices.end(), part.vertices.begin(), part.vertices.end());
            uvs.insert(uvs.end(), part.uvs.begin(), part.uvs.end());
            normals.insert(normals.end(), part.normals.begin(), part.normals.end());
            faces.insert(faces.end(), std::make_move_iterator(part.faces.begin()), std::make_move_iterator(part.faces.end()));
            part = Part{};
        }
    }

    // centre / scale / rotate (change_vectors, src/objectunion.cpp:291-315); applied only when a
    // centre is given (load_object_model, :434-436).  Vertex normals are left as read.
    if (tf.move_object && !vertices.empty()) {
        D3 mx = vertices[0], mn = vertices[0];
        for (const D3& v : vertices) {
            mx.x = std::max(v.x, mx.x);
            mx.y = std::max(v.y, mx.y);
            mx.z = std::max(v.z, mx.z);
            mn.x = std::min(v.x, mn.x);
            mn.y = std::min(v.y, mn.y);
            mn.z = std::min(v.z, mn.z);
        }
        D3 mid = {(mx.x + mn.x) / 2.0, (mx.y + mn.y) / 2.0, (mx.z + mn.z) / 2.0};
        double max_distance = 0;
        for (const D3& v : vertices) {
            double dx = v.x - mid.x, dy = v.y - mid.y, dz = v.z - mid.z;
            double d = std::sqrt(dx * dx + dy * dy + dz * dz);
            if (d > max_distance) max_distance = d;
        }
        for (D3& v : vertices) {
            D3 p = {v.x - mid.x, v.y - mid.y, v.z - mid.z};
            if (tf.scale_object) {
                p.x /= max_distance;
                p.y /= max_distance;
                p.z /= max_distance;
            }
            if (tf.rotate_object) p = rotate_yzx(p, tf.orientation[0], tf.orientation[1], tf.orientation[2]);
            if (tf.scale_object) {
                p.x = p.x * tf.size;
                p.y = p.y * tf.size;
                p.z = p.z * tf.size;
            }
            p.x += tf.center[0];
            p.y += tf.center[1];
            p.z += tf.center[2];
            v = p;
        }
    }

    first_triangle = static_cast<int64_t>(scene.tri_flags.size());
    // The faces, again a part per thread: every part emits its triangles into arrays of its own, which are appended to
    // the scene's in the order of the parts.
    struct Emitted {
        std::vector<double> tri_vertices, tri_uv, tri_normals;
        std::vector<uint8_t> tri_flags;
    };
    const int face_parts = faces.size() < 65536 && !std::getenv("PTK_OBJ_PARSE_PARTS") ? 1 : n_parts;
    std::vector<Emitted> emitted(static_cast<size_t>(face_parts));
    run_parts(face_parts, [&](int k, int n) {
        Emitted& out = emitted[static_cast<size_t>(k)];
        auto emit = [&](const std::string& face, int i1, int i2, int i3) -> bool {
            CornerRef c[3] = {parse_corner(nth_field(face, ' ', i1)), parse_corner(nth_field(face, ' ', i2)),
                              parse_corner(nth_field(face, ' ', i3))};
            if (!(c[0].v_ok && c[1].v_ok && c[2].v_ok)) return false;
            bool uv_ok = c[0].uv_ok && c[1].uv_ok && c[2].uv_ok;
            bool n_ok = c[0].n_ok && c[1].n_ok && c[2].n_ok;
            uint8_t flags = 0;
            for (int j = 0; j < 3; j++) {
                if (c[j].v < 0 || c[j].v >= static_cast<long>(vertices.size()))
                    throw LoadError(PTK_ERR_FORMAT, "Vertex index out of range in file '" + file_name + "'");
                const D3& v = vertices[static_cast<size_t>(c[j].v)];
                out.tri_vertices.insert(out.tri_vertices.end(), {v.x, v.y, v.z});
            }
            if (uv_ok) {
                for (int j = 0; j < 3; j++) {
                    if (c[j].uv < 0 || c[j].uv >= static_cast<long>(uvs.size()))
                        throw LoadError(PTK_ERR_FORMAT, "Texture index out of range in file '" + file_name + "'");
                    const D3& t = uvs[static_cast<size_t>(c[j].uv)];
                    out.tri_uv.insert(out.tri_uv.end(), {t.x, t.y});
                }
                flags |= PTK_TRI_HAS_UV;
            }
            else {
                out.tri_uv.insert(out.tri_uv.end(), 6, 0.0);
            }
            if (n_ok && smooth) {
                for (int j = 0; j < 3; j++) {
                    if (c[j].n < 0 || c[j].n >= static_cast<long>(normals.size()))
                        throw LoadError(PTK_ERR_FORMAT, "Normal index out of range in file '" + file_name + "'");
                    const D3& nn = normals[static_cast<size_t>(c[j].n)];
                    out.tri_normals.insert(out.tri_normals.end(), {nn.x, nn.y, nn.z});
                }
                flags |= PTK_TRI_SMOOTH;
            }
            else {
                out.tri_normals.insert(out.tri_normals.end(), 9, 0.0);
            }
            out.tri_flags.push_back(flags);
            return true;
        };
        const size_t begin = faces.size() * static_cast<size_t>(k) / static_cast<size_t>(n),
                     end = faces.size() * (static_cast<size_t>(k) + 1) / static_cast<size_t>(n);
        for (size_t f = begin; f < end; f++) {
            const std::string& face = faces[f];
            int spaces = count_char(face, ' ');
            if (spaces == 3) {
                emit(face, 1, 2, 3);
            }
            else if (spaces == 4) {
                // the second half of a quad follows the first's success (src/objectunion.cpp:404-419)
                if (emit(face, 1, 2, 3)) {
                    if (!emit(face, 1, 3, 4))
                        throw LoadError(PTK_ERR_FORMAT, "Malformed quad face in file '" + file_name + "'");
                }
            }
            // faces with any other number of spaces are silently dropped, as in the reference
        }
    });
    for (Emitted& part : emitted) {
        scene.tri_vertices.insert(scene.tri_vertices.end(), part.tri_vertices.begin(), part.tri_vertices.end());
        scene.tri_uv.insert(scene.tri_uv.end(), part.tri_uv.begin(), part.tri_uv.end());
        scene.tri_normals.insert(scene.tri_normals.end(), part.tri_normals.begin(), part.tri_normals.end());
        scene.tri_flags.insert(scene.tri_flags.end(), part.tri_flags.begin(), part.tri_flags.end());
        part = Emitted{};
    }
    n_triangles = static_cast<int64_t>(scene.tri_flags.size()) - first_triangle;
}

// -------------------------------------------------------------------------------------
// scene.json
// -------------------------------------------------------------------------------------
int find_index(const std::map<std::string, int>& store, const std::string& key, const std::string& key_type) {
    auto it = store.find(key);
    if (it == store.end()) throw LoadError(PTK_ERR_FORMAT, key_type + " entry '" + key + "' does not exist");
    return it->second;
}

void require_unique(const std::map<std::string, int>& store, const std::string& key, const std::string& key_type) {
    if (store.count(key)) throw LoadError(PTK_ERR_FORMAT, "Duplicate " + key_type + " entry '" + key + "' found");
}

void get_rotation_parameters(const Value& data, double yzx[3]) {
    require_field(data, "X");
    require_field(data, "Y");
    require_field(data, "Z");
    yzx[0] = data.at("Y").as_number();
    yzx[1] = data.at("Z").as_number();
    yzx[2] = data.at("X").as_number();
}

PtkLoadedScene* parse_scene(const std::string& text, const PtkSettings& settings, const std::string& asset_root) {
    Value root = ptkjson::parse(text);
    std::unique_ptr<PtkLoadedScene> scene(new PtkLoadedScene());
    std::map<std::string, int> map1d, map3d, media, materials, objects;
    const bool gamma = settings.use_gamma_correction != 0;

    require_field(root, "valuemaps");
    for (const Value& element : root.at("valuemaps").items) {
        require_field(element, "name");
        std::string name = element.at("name").as_string();
        require_field(element, "type");
        std::string type = element.at("type").as_string();
        if (type != "ValueMap1D" && type != "ValueMap3D")
            throw LoadError(PTK_ERR_FORMAT, "Found invalid type in ValueMap array: " + type);
        const bool is3 = type == "ValueMap3D";
        require_unique(is3 ? map3d : map1d, name, "valuemap");
        require_field(element, "parameters");
        const Value& parameters = element.at("parameters");
        PtkMap map{};
        if (parameters.contains("data")) {
            const Value& d = parameters.at("data");
            map.width = 1;
            map.height = 1;
            map.dim = is3 ? 3 : 1;
            map.offset = static_cast<int64_t>(scene->map_data.size());
            if (is3) {
                double c[3];
                get_vec3_param(parameters, "data", c);
                for (int i = 0; i < 3; i++) scene->map_data.push_back(gamma ? srgb_to_linear(c[i]) : c[i]);
            }
            else {
                if (d.kind != Value::Array || d.size() != 1)
                    throw LoadError(PTK_ERR_FORMAT, "Data array " + d.dump() +
                                                        " should contain 1 element, but contains " +
                                                        std::to_string(d.size()));
                scene->map_data.push_back(d.items[0].as_number());
            }
        }
        else if (parameters.contains("file")) {
            load_map_file(parameters.at("file").as_string(), asset_root, is3 && gamma, is3 ? 3 : 1, *scene, map);
        }
        else {
            throw LoadError(PTK_ERR_FORMAT, "ValueMap must contain either 'data' or 'file'");
        }
        (is3 ? map3d : map1d)[name] = static_cast<int>(scene->maps.size());
        scene->maps.push_back(map);
    }

    require_field(root, "media");
    for (const Value& element : root.at("media").items) {
        require_field(element, "name");
        std::string name = element.at("name").as_string();
        require_field(element, "type");
        std::string type = element.at("type").as_string();
        if (type != "Medium") throw LoadError(PTK_ERR_FORMAT, "Found invalid type in medium array: " + type);
        require_unique(media, name, "medium");
        require_field(element, "parameters");
        const Value& p = element.at("parameters");
        require_field(p, "scattering_albedo");
        require_field(p, "absorption_albedo");
        require_field(p, "emission_coefficient");
        require_field(p, "refractive_index");
        PtkMedium m{};
        get_vec3_param(p, "scattering_albedo", m.scattering);
        get_vec3_param(p, "absorption_albedo", m.absorption);
        get_vec3_param(p, "emission_coefficient", m.emission);
        m.refractive_index = p.at("refractive_index").as_number();
        require_field(element, "subtype");
        std::string subtype = element.at("subtype").as_string();
        if (subtype == "BeersLawMedium")
            m.type = PTK_MEDIUM_BEERS_LAW;
        else if (subtype == "HomogenousScatteringMedium")
            m.type = PTK_MEDIUM_HOMOGENEOUS_SCATTERING;
        else
            throw LoadError(PTK_ERR_FORMAT, subtype + " is not a valid medium type");
        media[name] = static_cast<int>(scene->media.size());
        scene->media.push_back(m);
    }

    if (root.contains("materials")) {
        for (const Value& element : root.at("materials").items) {
            require_field(element, "name");
            std::string name = element.at("name").as_string();
            require_field(element, "type");
            std::string type = element.at("type").as_string();
            if (type != "Material") throw LoadError(PTK_ERR_FORMAT, "Found invalid type in material array: " + type);
            require_unique(materials, name, "material");
            require_field(element, "parameters");
            const Value& p = element.at("parameters");
            PtkMaterial m{};
            m.is_light_source = 0;
            m.is_dielectric = 1;
            m.surface_refractive_index = 1.0;
            m.extinction_coefficient = 2.0;
            m.albedo_map = m.emission_color_map = m.light_intensity_map = m.roughness_map = m.transparency_map = -1;
            m.internal_medium = m.external_medium = -1;
            if (p.contains("albedo_map")) m.albedo_map = find_index(map3d, p.at("albedo_map").as_string(), "ValueMap3D");
            if (p.contains("surface_refractive_index"))
                m.surface_refractive_index = p.at("surface_refractive_index").as_number();
            if (p.contains("extinction_coefficient"))
                m.extinction_coefficient = p.at("extinction_coefficient").as_number();
            if (p.contains("emission_color_map"))
                m.emission_color_map = find_index(map3d, p.at("emission_color_map").as_string(), "ValueMap3D");
            if (p.contains("light_intensity_map"))
                m.light_intensity_map = find_index(map1d, p.at("light_intensity_map").as_string(), "ValueMap1D");
            if (p.contains("is_dielectric")) m.is_dielectric = p.at("is_dielectric").as_bool() ? 1 : 0;
            if (p.contains("roughness_map"))
                m.roughness_map = find_index(map1d, p.at("roughness_map").as_string(), "ValueMap1D");
            if (p.contains("is_light_source")) m.is_light_source = p.at("is_light_source").as_bool() ? 1 : 0;
            if (p.contains("internal_medium"))
                m.internal_medium = find_index(media, p.at("internal_medium").as_string(), "internal_medium");
            if (p.contains("external_medium"))
                m.external_medium = find_index(media, p.at("external_medium").as_string(), "external_medium");
            if (p.contains("transparency_map"))
                m.transparency_map = find_index(map1d, p.at("transparency_map").as_string(), "ValueMap1D");
            require_field(element, "subtype");
            std::string subtype = element.at("subtype").as_string();
            static const char* names[PTK_MAT_COUNT] = {"Diffuse",
                                                       "Reflective",
                                                       "Transparent",
                                                       "Glossy",
                                                       "MetallicMicrofacet",
                                                       "ReflectiveMicrofacet",
                                                       "TransparentMicrofacet"};
            m.type = -1;
            for (int i = 0; i < PTK_MAT_COUNT; i++)
                if (subtype == names[i]) m.type = i;
            if (m.type < 0) throw LoadError(PTK_ERR_FORMAT, subtype + " is not a valid material type");
            materials[name] = static_cast<int>(scene->materials.size());
            scene->materials.push_back(m);
        }
    }

    if (root.contains("objects")) {
        for (const Value& element : root.at("objects").items) {
            require_field(element, "name");
            std::string name = element.at("name").as_string();
            require_field(element, "type");
            std::string type = element.at("type").as_string();
            if (type != "Object") throw LoadError(PTK_ERR_FORMAT, "Found invalid type in object array: " + type);
            require_unique(objects, name, "object");
            require_field(element, "subtype");
            std::string subtype = element.at("subtype").as_string();
            require_field(element, "parameters");
            const Value& p = element.at("parameters");
            require_field(p, "material");
            PtkObject o{};
            o.material = find_index(materials, p.at("material").as_string(), "Material");
            if (subtype == "Sphere") {
                require_field(p, "position");
                require_field(p, "radius");
                o.type = PTK_OBJ_SPHERE;
                get_vec3_param(p, "position", o.position);
                o.radius = p.at("radius").as_number();
            }
            else if (subtype == "Plane" || subtype == "Rectangle") {
                require_field(p, "position");
                require_field(p, "v1");
                require_field(p, "v2");
                o.type = subtype == "Plane" ? PTK_OBJ_PLANE : PTK_OBJ_RECTANGLE;
                if (o.type == PTK_OBJ_RECTANGLE) {
                    require_field(p, "L1");
                    require_field(p, "L2");
                    o.L1 = p.at("L1").as_number();
                    o.L2 = p.at("L2").as_number();
                }
                get_vec3_param(p, "position", o.position);
                get_vec3_param(p, "v1", o.v1);
                get_vec3_param(p, "v2", o.v2);
            }
            else if (subtype == "Triangle") {
                require_field(p, "p1");
                require_field(p, "p2");
                require_field(p, "p3");
                o.type = PTK_OBJ_TRIANGLE;
                get_vec3_param(p, "p1", o.position);
                get_vec3_param(p, "p2", o.v1);
                get_vec3_param(p, "p3", o.v2);
            }
            else if (subtype == "ObjectUnion") {
                require_field(p, "file");
                o.type = PTK_OBJ_MESH;
                bool smooth = p.contains("enable_smooth_shading") ? p.at("enable_smooth_shading").as_bool() : false;
                ObjTransform tf;
                tf.move_object = p.contains("center");
                tf.scale_object = p.contains("size");
                tf.rotate_object = p.contains("orientation");
                if (tf.move_object) get_vec3_param(p, "center", tf.center);
                if (tf.scale_object) tf.size = p.at("size").as_number();
                if (tf.rotate_object) get_rotation_parameters(p.at("orientation"), tf.orientation);
                load_obj(p.at("file").as_string(), asset_root, smooth, tf, *scene, o.first_triangle, o.n_triangles);
            }
            else {
                throw LoadError(PTK_ERR_FORMAT, subtype + " is not a valid object type");
            }
            objects[name] = static_cast<int>(scene->objects.size());
            scene->objects.push_back(o);
            scene->object_names.push_back(name);
        }
    }

    PtkSceneDesc& d = scene->desc;
    d.settings = settings;
    require_field(root, "background_medium");
    d.background_medium = find_index(media, root.at("background_medium").as_string(), "Medium");

    require_field(root, "camera");
    const Value& cam = root.at("camera");
    require_field(cam, "camera_position");
    get_vec3_param(cam, "camera_position", d.camera.position);
    require_field(cam, "orientation");
    double yzx[3];
    get_rotation_parameters(cam.at("orientation"), yzx);
    d.camera.orientation_x = yzx[2];
    d.camera.orientation_y = yzx[0];
    d.camera.orientation_z = yzx[1];
    require_field(cam, "camera_width");
    d.camera.camera_width = cam.at("camera_width").as_number();
    require_field(cam, "focal_length");
    d.camera.focal_length = cam.at("focal_length").as_number();
    d.camera.dof_mode = PTK_DOF_NONE;
    if (cam.contains("depth_of_field_mode")) {
        std::string mode = cam.at("depth_of_field_mode").as_string();
        require_field(cam, "focus_distance");
        require_field(cam, "aperture_size");
        d.camera.focus_distance = cam.at("focus_distance").as_number();
        d.camera.aperture_size = cam.at("aperture_size").as_number();
        if (mode == "circle")
            d.camera.dof_mode = PTK_DOF_CIRCLE;
        else if (mode == "square")
            d.camera.dof_mode = PTK_DOF_SQUARE;
        else if (mode != "none")
            throw LoadError(PTK_ERR_FORMAT, "Invalid 'depth_of_field_mode' " + mode + " in scene definition.");
    }
    d.background_color[0] = d.background_color[1] = d.background_color[2] = 0.0;
    if (root.contains("background_color")) get_vec3_param(root, "background_color", d.background_color);

    d.n_objects = static_cast<int32_t>(scene->objects.size());
    d.n_materials = static_cast<int32_t>(scene->materials.size());
    d.n_media = static_cast<int32_t>(scene->media.size());
    d.n_maps = static_cast<int32_t>(scene->maps.size());
    d.objects = scene->objects.data();
    d.materials = scene->materials.data();
    d.media = scene->media.data();
    d.maps = scene->maps.data();
    d.map_data = scene->map_data.data();
    d.n_map_values = static_cast<int64_t>(scene->map_data.size());
    d.n_triangles = static_cast<int64_t>(scene->tri_flags.size());
    d.tri_vertices = scene->tri_vertices.data();
    d.tri_uv = scene->tri_uv.data();
    d.tri_normals = scene->tri_normals.data();
    d.tri_flags = scene->tri_flags.data();
    return scene.release();
}

template<typename F>
int guarded(F&& body) {
    try {
        body();
        return PTK_OK;
    }
    catch (const LoadError& e) {
        return fail(e.code, e.what());
    }
    catch (const std::exception& e) {
        return fail(PTK_ERR_FORMAT, e.what());
    }
}

} // namespace

extern "C" {

const char* ptk_host_last_error(void) {
    return g_error.c_str();
}

void ptk_host_default_settings(PtkSettings* out) {
    if (out) default_settings(*out);
}

int ptk_host_load_settings(const char* path, PtkSettings* out) {
    if (!path || !out) return fail(PTK_ERR_INVALID, "ptk_host_load_settings: null argument");
    std::string text;
    if (!read_text_file(path, text)) return fail(PTK_ERR_IO, "Could not open file '" + std::string(path) + "'");
    return guarded([&] { parse_settings(text, *out); });
}

int ptk_host_load_scene_text(const char* text, const PtkSettings* settings, const char* asset_root,
                             PtkLoadedScene** out) {
    if (!text || !settings || !out) return fail(PTK_ERR_INVALID, "ptk_host_load_scene_text: null argument");
    *out = nullptr;
    return guarded([&] { *out = parse_scene(text, *settings, asset_root ? asset_root : ""); });
}

int ptk_host_load_scene(const char* path, const PtkSettings* settings, const char* asset_root, PtkLoadedScene** out) {
    if (!path || !settings || !out) return fail(PTK_ERR_INVALID, "ptk_host_load_scene: null argument");
    std::string text;
    if (!read_text_file(path, text)) return fail(PTK_ERR_IO, "Could not open file '" + std::string(path) + "'");
    return ptk_host_load_scene_text(text.c_str(), settings, asset_root, out);
}

const PtkSceneDesc* ptk_host_scene_desc(const PtkLoadedScene* scene) {
    return scene ? &scene->desc : nullptr;
}

const char* ptk_host_object_name(const PtkLoadedScene* scene, int32_t index) {
    if (!scene || index < 0 || index >= static_cast<int32_t>(scene->object_names.size())) return nullptr;
    return scene->object_names[static_cast<size_t>(index)].c_str();
}

void ptk_host_free_scene(PtkLoadedScene* scene) {
    delete scene;
}

int ptk_host_load_denoising(const char* path, const PtkSettings* settings, PtkDenoiseTask** tasks, int32_t* n_tasks) {
    if (!path || !settings || !tasks || !n_tasks) return fail(PTK_ERR_INVALID, "ptk_host_load_denoising: null argument");
    *tasks = nullptr;
    *n_tasks = 0;
    std::string text;
    if (!read_text_file(path, text)) return fail(PTK_ERR_IO, "Could not open file '" + std::string(path) + "'");
    std::vector<PtkDenoiseTask> pipeline;
    int rc = guarded([&] {
        Value data = ptkjson::parse(text);
        require_field(data, "pipeline");
        int min_dim = std::min(settings->width, settings->height);
        for (const Value& element : data.at("pipeline").items) {
            require_field(element, "mode");
            require_field(element, "parameters");
            std::string mode = element.at("mode").as_string();
            const Value& p = element.at("parameters");
            if (mode == "skip") continue;
            PtkDenoiseTask task{};
            if (mode == "atrous") {
                require_field(p, "iterations");
                require_field(p, "sigma_rt");
                require_field(p, "sigma_x");
                require_field(p, "sigma_n");
                task.mode = PTK_DENOISE_ATROUS;
                task.iterations = static_cast<int32_t>(p.at("iterations").as_number());
                task.sigma_rt = p.at("sigma_rt").as_number();
                task.sigma_x = p.at("sigma_x").as_number();
                task.sigma_n = p.at("sigma_n").as_number();
                if (std::pow(2, task.iterations) >= static_cast<double>(min_dim))
                    throw LoadError(PTK_ERR_FORMAT, "Too many denoising iterations for this image size.");
            }
            else if (mode == "median") {
                require_field(p, "kernel_size");
                require_field(p, "threshold");
                task.mode = PTK_DENOISE_MEDIAN;
                task.kernel_size = static_cast<int32_t>(p.at("kernel_size").as_number());
                task.threshold = p.at("threshold").as_number();
                if (task.kernel_size > min_dim)
                    throw LoadError(PTK_ERR_FORMAT, "Median filter kernel size is too large for this image size.");
                if (task.threshold < 0) throw LoadError(PTK_ERR_FORMAT, "Median filter threshold cannot be negative.");
            }
            else {
                throw LoadError(PTK_ERR_FORMAT, "'" + mode + "' is not a valid denoising mode.");
            }
            pipeline.push_back(task);
        }
    });
    if (rc != PTK_OK) return rc;
    *n_tasks = static_cast<int32_t>(pipeline.size());
    if (!pipeline.empty()) {
        *tasks = static_cast<PtkDenoiseTask*>(std::malloc(pipeline.size() * sizeof(PtkDenoiseTask)));
        if (!*tasks) return fail(PTK_ERR_MEMORY, "out of memory");
        std::memcpy(*tasks, pipeline.data(), pipeline.size() * sizeof(PtkDenoiseTask));
    }
    return PTK_OK;
}

double* ptk_host_map_buffer(const char* path, size_t file_size, int clear_content, int* fd_out) {
    int fd = open(path, O_RDWR | O_CREAT, S_IRUSR | S_IWUSR);
    if (fd == -1) {
        fail(PTK_ERR_IO, std::string("Error opening file: ") + std::strerror(errno));
        return nullptr;
    }
    if (clear_content && ftruncate(fd, 0) == -1) {
        fail(PTK_ERR_IO, std::string("Error clearing temporary file: ") + std::strerror(errno));
        close(fd);
        return nullptr;
    }
    if (ftruncate(fd, static_cast<off_t>(file_size)) == -1) {
        fail(PTK_ERR_IO, std::string("Error setting file size: ") + std::strerror(errno));
        close(fd);
        return nullptr;
    }
    void* mapping = mmap(nullptr, file_size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (mapping == MAP_FAILED) {
        fail(PTK_ERR_IO, std::string("Error mapping file: ") + std::strerror(errno));
        close(fd);
        return nullptr;
    }
    if (fd_out) *fd_out = fd;
    return static_cast<double*>(mapping);
}

void ptk_host_unmap_buffer(double* mapping, size_t file_size, int fd) {
    if (mapping && munmap(mapping, file_size) == -1) perror("Error un-mapping the file");
    if (fd >= 0) close(fd);
}

int ptk_host_write_map(const char* path, int32_t width, int32_t height, int32_t dim, const double* data) {
    if (!path || !data || width < 1 || height < 1 || dim < 1) return fail(PTK_ERR_INVALID, "ptk_host_write_map: bad argument");
    FILE* f = std::fopen(path, "wb");
    if (!f) return fail(PTK_ERR_IO, "Could not open file '" + std::string(path) + "'");
    double header[3] = {static_cast<double>(width), static_cast<double>(height), static_cast<double>(dim)};
    size_t n = static_cast<size_t>(width) * static_cast<size_t>(height) * static_cast<size_t>(dim);
    bool ok = std::fwrite(header, sizeof(double), 3, f) == 3 && std::fwrite(data, sizeof(double), n, f) == n;
    std::fclose(f);
    return ok ? PTK_OK : fail(PTK_ERR_IO, "Error writing file '" + std::string(path) + "'");
}

} // extern "C"
