// TEST INFRASTRUCTURE -- CPU restatement of the reference path tracer's hot path.
// See pt_oracle.h for what this is for and who may use it.
//
// Every function names the reference code it restates (paths relative to the reference
// repository).  Floating-point expressions keep the reference's operation order so that,
// with the reference's random streams, results are bit-identical to a single-threaded
// reference run; the data model is different (flat tables + switch instead of virtual
// classes).
#include "pt_oracle.h"

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <random>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_error;
const double EPSILON = 0.000001;                                  // src/constants.cpp:22
const double INF = std::numeric_limits<double>::infinity();       // max_ray_distance, :23

// ---------------------------------------------------------------------------------
// vec3 (include/vec3.h): component-wise ops, dot summed left to right
// ---------------------------------------------------------------------------------
struct V3 {
    double x = 0, y = 0, z = 0;
    V3() {}
    V3(double a, double b, double c) : x(a), y(b), z(c) {}
    explicit V3(double a) : x(a), y(a), z(a) {}
    explicit V3(const double* p) : x(p[0]), y(p[1]), z(p[2]) {}
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    double& at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator-(const V3& a) { return V3(-a.x, -a.y, -a.z); }
inline V3 operator+(const V3& a, const V3& b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(const V3& a, const V3& b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(const V3& a, const V3& b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator*(const V3& a, double s) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(double s, const V3& a) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator/(const V3& a, double s) { return V3(a.x / s, a.y / s, a.z / s); }
inline V3& operator+=(V3& a, const V3& b) {
    a.x += b.x;
    a.y += b.y;
    a.z += b.z;
    return a;
}
inline V3& operator*=(V3& a, const V3& b) {
    a.x *= b.x;
    a.y *= b.y;
    a.z *= b.z;
    return a;
}
inline V3& operator*=(V3& a, double s) {
    a.x *= s;
    a.y *= s;
    a.z *= s;
    return a;
}
inline V3& operator/=(V3& a, double s) {
    a.x /= s;
    a.y /= s;
    a.z /= s;
    return a;
}
inline double dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(const V3& a, const V3& b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline double len2(const V3& a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
inline double len(const V3& a) { return std::sqrt(len2(a)); }
inline V3 normalize(const V3& a) { return a / len(a); }
inline double vmax(const V3& a) { return std::max(std::max(a.x, a.y), a.z); }
inline V3 vexp(const V3& a) { return V3(std::exp(a.x), std::exp(a.y), std::exp(a.z)); }
inline int argmax_abs(const V3& d) { // argmax(abs(v)), include/vec3.h:156-165
    double v[3] = {std::abs(d.x), std::abs(d.y), std::abs(d.z)};
    int m = 0;
    if (v[1] > v[m]) m = 1;
    if (v[2] > v[m]) m = 2;
    return m;
}
inline V3 permute(const V3& v, int a, int b, int c) { return V3(v[a], v[b], v[c]); }

// ---------------------------------------------------------------------------------
// random numbers
// ---------------------------------------------------------------------------------
const uint64_t GOLDEN = 0x9E3779B97F4A7C15ull;
const uint64_t FORK_SALT = 0xD1B54A32D192ED03ull;

inline uint64_t mix64(uint64_t z) {
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}
inline uint64_t rng_key(uint64_t seed, uint32_t pixel, uint32_t sample) {
    return mix64(mix64(seed + GOLDEN) ^ ((static_cast<uint64_t>(pixel) << 32) | sample));
}
inline double rng_uniform(uint64_t key, uint32_t n) {
    uint64_t bits = mix64(key + (static_cast<uint64_t>(n) + 1) * GOLDEN);
    return static_cast<double>(bits >> 11) * 0x1.0p-53;
}
inline uint64_t rng_fork(uint64_t key, uint32_t n) {
    return mix64(key ^ mix64(static_cast<uint64_t>(n) + FORK_SALT));
}

// The three generator front ends of the reference (random_uniform / random_normal /
// rand(), src/utils.cpp:19-30,120-121) behind one object.
struct Rng {
    int mode = ORACLE_RNG_COUNTER;
    // counter mode
    uint64_t key = 0;
    uint32_t ctr = 0;
    uint64_t outer_key = 0;
    uint32_t outer_ctr = 0;
    // reference-stream mode
    std::minstd_rand ugen, ngen;
    std::uniform_real_distribution<double> udist{0, 1};
    std::normal_distribution<double> ndist{0, 1};

    void start_path(uint64_t seed, uint32_t pixel, uint32_t sample) {
        if (mode == ORACLE_RNG_COUNTER) {
            key = rng_key(seed, pixel, sample);
            ctr = 0;
        }
    }
    double u01() {
        if (mode == ORACLE_RNG_COUNTER) return rng_uniform(key, ctr++);
        return udist(ugen);
    }
    double uniform(double low, double high) { return (high - low) * u01() + low; } // random_uniform
    int integer(int low, int high) { return static_cast<int>(uniform(low, high)); } // random_int
    double normal() {                                                                // random_normal
        if (mode == ORACLE_RNG_COUNTER) {
            double u1 = u01();
            double u2 = u01();
            return std::sqrt(-2.0 * std::log(1.0 - u1)) * std::cos(2.0 * M_PI * u2);
        }
        return ndist(ngen);
    }
    double crand() { // (double) rand() / RAND_MAX
        if (mode == ORACLE_RNG_COUNTER) return u01();
        return static_cast<double>(rand()) / (RAND_MAX);
    }
    // The visibility march draws from a forked key so that the path's own counter does
    // not depend on how many surfaces the shadow ray crosses.
    void fork_begin() {
        if (mode != ORACLE_RNG_COUNTER) return;
        outer_key = key;
        outer_ctr = ctr + 1;
        key = rng_fork(key, ctr);
        ctr = 0;
    }
    void fork_end() {
        if (mode != ORACLE_RNG_COUNTER) return;
        key = outer_key;
        ctr = outer_ctr;
    }
};

// ---------------------------------------------------------------------------------
// scene model
// ---------------------------------------------------------------------------------
enum RayType { DIFFUSE = 0, REFLECTED = 1, TRANSMITTED = 2 }; // include/utils.h:14

struct Ray { // include/utils.h:32-63
    double t_max = INF;
    double Sx = 0, Sy = 0, Sz = 0;
    int kx = 0, ky = 0, kz = 0;
    int type = DIFFUSE;
    V3 origin, dir;
    void prepare() {
        kz = argmax_abs(dir);
        kx = kz + 1;
        if (kx == 3) kx = 0;
        ky = kx + 1;
        if (ky == 3) ky = 0;
        V3 d = permute(dir, kx, ky, kz);
        Sx = -d.x / d.z;
        Sy = -d.y / d.z;
        Sz = 1.0 / d.z;
    }
};

struct Hit { // include/utils.h:22-30
    int object = -1;
    int primitive = -1;
    bool outside = false;
    double distance = INF;
    V3 point, incident, normal;
};

struct Map {
    int width = 1, height = 1, dim = 1;
    const double* data = nullptr;
};

struct Medium { // include/medium.h
    int type = PTK_MEDIUM_BEERS_LAW;
    double ior = 1;
    V3 absorption, scattering, extinction, emission;
};

struct Material { // include/materials.h:55-82 after Material::Material (src/materials.cpp:3-43)
    int type = PTK_MAT_DIFFUSE;
    bool is_light = false, is_dielectric = true;
    double n = 1, k = 0;
    int albedo = -1, emission = -1, intensity = -1, roughness = -1, transparency = -1;
    int internal_medium = -1, external_medium = -1;
};

struct Tri { // Triangle (include/objects.h:97-139) after its constructor
    V3 p1, p2, p3, v1, v2, normal;
    double x1, y1, x2, y2, x3, y3, det_T, area;
    V3 uv1, uv2, uv3, n1, n2, n3;
    bool smooth = false;
    int64_t face_index = 0;
};

struct BvhNode { // BVH::Node (include/bvh.h:46-63)
    V3 lo, hi;
    bool leaf = false;
    int child[2] = {-1, -1};
    std::vector<int> prims; // leaf: primitive ids in the leaf array's order
};

struct Mesh { // ObjectUnion (include/objectunion.h:21-49)
    std::vector<Tri> tris; // in the order the BVH root left them: index = primitive_ID
    std::vector<BvhNode> nodes;
    double area = 0;
    size_t n_lights = 0;
    std::vector<double> cumulative_area;
    std::vector<int> light_index;
};

struct Object {
    int type = PTK_OBJ_SPHERE;
    int material = 0;
    double area = 0;
    V3 position;
    double radius = 0;
    V3 v1, v2, normal;
    double L1 = 0, L2 = 0;
    Tri tri;
    int mesh = -1;
};

} // namespace

struct OracleScene {
    PtkSettings settings{};
    std::vector<double> map_data;
    std::vector<Map> maps;
    std::vector<Medium> media;
    std::vector<Material> materials;
    std::vector<Object> objects;
    std::vector<Mesh> meshes;
    int background_medium = 0;
    V3 background_color;
    // Camera (include/camera.h)
    V3 cam_position, cam_view, cam_screen_x, cam_screen_y;
    double cam_width = 0, cam_screen_height = 0, cam_focal = 0, cam_aperture = 0, cam_focus = 0;
    int cam_dof = PTK_DOF_NONE;
    std::mutex stats_mutex;
    OracleStats stats{};
};

namespace {

struct Ctx {
    const OracleScene* s;
    Rng rng;
    OracleStats stats{};
    bool shadow = false;
};

// ---------------------------------------------------------------------------------
// src/utils.cpp
// ---------------------------------------------------------------------------------
double pos_fmod(double a, double b) { return std::fmod((std::fmod(a, b) + b), b); } // :32-34

bool solve_quadratic(double b, double c, double& distance) { // :52-71
    double discriminant = b * b - 4.0 * c;
    if (discriminant <= 0) return false;
    double root = std::sqrt(discriminant);
    double minimum_solution = -1.0 / 2.0 * (b + root);
    if (minimum_solution > EPSILON) {
        distance = minimum_solution;
        return true;
    }
    double maximum_solution = -1.0 / 2.0 * (b - root);
    if (maximum_solution > EPSILON) {
        distance = maximum_solution;
        return true;
    }
    return false;
}

V3 sample_spherical(Rng& rng) { // :73-80
    double r1 = rng.normal();
    double r2 = rng.normal();
    double r3 = rng.normal();
    return normalize(V3(r1, r2, r3));
}

void perpendicular_vectors(const V3& z_hat, V3& x_hat, V3& y_hat) { // :90-100
    V3 non_parallel(1.0, 0.0, 0.0);
    if (std::abs(dot(non_parallel, z_hat)) == 1.0) non_parallel = V3(0.0, 1.0, 0.0);
    x_hat = cross(z_hat, non_parallel);
    x_hat = normalize(x_hat);
    y_hat = cross(z_hat, x_hat);
    y_hat = normalize(y_hat);
}

V3 sample_cosine_hemisphere(const V3& normal, Rng& rng) { // :115-126
    V3 x_hat, y_hat;
    perpendicular_vectors(normal, x_hat, y_hat);
    double theta = rng.crand() * 2 * M_PI;
    double radius = std::sqrt(rng.crand());
    double x = std::cos(theta) * radius;
    double y = std::sin(theta) * radius;
    double z = std::sqrt(1 - x * x - y * y);
    return x_hat * x + y_hat * y + normal * z;
}

V3 reflect_vector(const V3& direction, const V3& normal) { // :128-130
    return direction - normal * 2.0 * dot(normal, direction);
}

V3 refract_vector(const V3& incident, const V3& normal, double eta) { // :132-141
    double cos_incident = dot(normal, incident);
    double k = 1 - eta * eta * (1 - cos_incident * cos_incident);
    if (k < 0) return V3(0, 0, 0);
    V3 perpendicular = incident - normal * cos_incident;
    return normal * std::sqrt(k) + perpendicular * eta;
}

V3 rotate(const V3& p, double alpha, double beta, double gamma) { // :143-173
    double alpha_rad = alpha * M_PI / 180;
    double beta_rad = beta * M_PI / 180;
    double gamma_rad = gamma * M_PI / 180;
    double x = p.x, y = p.y, z = p.z;
    double cos_alpha = std::cos(alpha_rad), sin_alpha = std::sin(alpha_rad);
    double cos_beta = std::cos(beta_rad), sin_beta = std::sin(beta_rad);
    double cos_gamma = std::cos(gamma_rad), sin_gamma = std::sin(gamma_rad);
    double e1 = x * (cos_alpha * cos_beta) + y * (sin_alpha * sin_gamma - cos_alpha * cos_gamma * sin_beta) +
                z * (cos_gamma * sin_alpha + cos_alpha * sin_beta * sin_gamma);
    double e2 = x * (sin_beta) + y * (cos_beta * cos_gamma) + z * (-cos_beta * sin_gamma);
    double e3 = x * (-cos_beta * sin_alpha) + y * (cos_alpha * sin_gamma + cos_gamma * sin_alpha * sin_beta) +
                z * (cos_alpha * cos_gamma - sin_alpha * sin_beta * sin_gamma);
    return V3(e1, e2, e3);
}

double fresnel_dielectric(double cos_incident, double n1, double n2) { // :175-189
    double sin_incident = std::sqrt(1 - cos_incident * cos_incident);
    double cos_transmitted = std::sqrt(1 - std::pow(n1 / n2 * sin_incident, 2));
    if (std::isnan(cos_transmitted)) return 0.0;
    double n1_cos_incident = n1 * cos_incident;
    double n2_cos_transmitted = n2 * cos_transmitted;
    double n1_cos_transmitted = n1 * cos_transmitted;
    double n2_cos_incident = n2 * cos_incident;
    double R_s = std::pow((n1_cos_incident - n2_cos_transmitted) / (n1_cos_incident + n2_cos_transmitted), 2);
    double R_p = std::pow((n1_cos_transmitted - n2_cos_incident) / (n1_cos_transmitted + n2_cos_incident), 2);
    return 0.5 * (R_s + R_p);
}

double fresnel_conductor(double cos_theta_real, double n1, double k1, double n2, double k2) { // :191-223
    typedef std::complex<double> C;
    C cos_theta(cos_theta_real, 0);
    double eta, k;
    C one(1, 0);
    if (k1 == 0) {
        eta = n2 / n1;
        k = k2 / n1;
    }
    else {
        eta = n1 / n2;
        k = k1 / n2;
        C sin_theta = std::sqrt(one - cos_theta * cos_theta);
        C n_tilde(n1, k1);
        C sin_theta_T = n_tilde / n2 * sin_theta;
        C sin_theta_T2 = sin_theta_T * sin_theta_T;
        cos_theta = std::sqrt(one - sin_theta_T2);
    }
    C cos_theta2 = cos_theta * cos_theta;
    C sin_theta2 = one - cos_theta2;
    C f0 = std::sqrt(std::pow(eta * eta - k * k - sin_theta2, 2) + 4.0 * eta * eta * k * k);
    C a2b2 = f0;
    C a = std::sqrt(0.5 * f0 + eta * eta - k * k - sin_theta2);
    C f1 = a2b2 + cos_theta2;
    C f2 = 2.0 * a * cos_theta;
    C f3 = cos_theta2 * a2b2 + sin_theta2 * sin_theta2;
    C f4 = f2 * sin_theta2;
    C R_s = (f1 - f2) / (f1 + f2);
    C R_p = R_s * (f3 - f4) / (f3 + f4);
    return 0.5 * std::real(R_p + R_s);
}

double schlick_fresnel(double cos_incident, double n1, double n2) { // :225-234
    double R_0_sqrt = (n1 - n2) / (n1 + n2);
    double R_0 = R_0_sqrt * R_0_sqrt;
    double x = 1 - cos_incident;
    double fresnel = R_0 + (1 - R_0) * (x * x * x * x * x);
    if (std::isnan(fresnel) || fresnel < 0 || fresnel > 1) return 0;
    return fresnel;
}

double fresnel_multiplier(double cos_incident, double n1, double k1, double n2, double k2, bool is_dielectric) { // :236-243
    if (is_dielectric || (k1 == 0 && k2 == 0)) return fresnel_dielectric(cos_incident, n1, n2);
    return fresnel_conductor(cos_incident, n1, k1, n2, k2);
}

// ---------------------------------------------------------------------------------
// textures (src/valuemap.cpp:34-52); map id -1 is the constant-zero default map
// (src/materials.cpp:4-22)
// ---------------------------------------------------------------------------------
size_t texel_index(const Map& m, double u, double v) {
    size_t u_idx = static_cast<size_t>(static_cast<double>(m.width) * pos_fmod(u, 1.0));
    size_t v_idx = static_cast<size_t>(static_cast<double>(m.height) * pos_fmod(1 - v, 1.0));
    return v_idx * static_cast<size_t>(m.width) + u_idx;
}
double map1(const OracleScene& s, int id, double u, double v) {
    if (id < 0) return 0;
    if (std::isnan(u) || std::isnan(v)) return 0;
    const Map& m = s.maps[static_cast<size_t>(id)];
    return m.data[texel_index(m, u, v)];
}
V3 map3(const OracleScene& s, int id, double u, double v) {
    if (id < 0) return V3(0, 0, 0);
    if (std::isnan(u) || std::isnan(v)) return V3(0, 0, 0);
    const Map& m = s.maps[static_cast<size_t>(id)];
    size_t start = 3 * texel_index(m, u, v);
    return V3(m.data[start], m.data[start + 1], m.data[start + 2]);
}

// ---------------------------------------------------------------------------------
// triangles (src/objects.cpp:269-412)
// ---------------------------------------------------------------------------------
Tri make_triangle(const V3& p1, const V3& p2, const V3& p3) { // Triangle::Triangle :269-299
    Tri t;
    t.p1 = p1;
    t.p2 = p2;
    t.p3 = p3;
    t.v1 = p2 - p1;
    t.v2 = p3 - p1;
    t.normal = normalize(cross(t.v1, t.v2));
    t.v1 = normalize(t.v1);
    t.v2 = normalize(cross(t.normal, t.v1));
    t.x1 = dot(p1, t.v1);
    t.y1 = dot(p1, t.v2);
    t.x2 = dot(p2, t.v1);
    t.y2 = dot(p2, t.v2);
    t.x3 = dot(p3, t.v1);
    t.y3 = dot(p3, t.v2);
    t.det_T = (t.y2 - t.y3) * (t.x1 - t.x3) + (t.x3 - t.x2) * (t.y1 - t.y3);
    t.area = 0.5 * std::abs(t.x1 * (t.y2 - t.y3) + t.x2 * (t.y3 - t.y1) + t.x3 * (t.y1 - t.y2));
    t.n1 = t.n2 = t.n3 = t.normal;
    return t;
}

V3 tri_barycentric(const Tri& t, const V3& point) { // compute_barycentric :346-353
    double x = dot(point, t.v1);
    double y = dot(point, t.v2);
    double lambda1 = ((t.y2 - t.y3) * (x - t.x3) + (t.x3 - t.x2) * (y - t.y3)) / t.det_T;
    double lambda2 = ((t.y3 - t.y1) * (x - t.x3) + (t.x1 - t.x3) * (y - t.y3)) / t.det_T;
    return V3(lambda1, lambda2, 1.0 - lambda1 - lambda2);
}

V3 tri_uv(const Tri& t, const V3& point) { // get_UV :355-358
    V3 b = tri_barycentric(t, point);
    return t.uv1 * b.x + t.uv2 * b.y + t.uv3 * b.z;
}

V3 tri_normal(const Tri& t, const V3& point) { // get_normal_vector :334-344
    if (t.smooth) {
        V3 b = tri_barycentric(t, point);
        return normalize(t.n1 * b.x + t.n2 * b.y + t.n3 * b.z);
    }
    return t.normal;
}

bool tri_intersect(const Tri& t, const Ray& ray, double& distance) { // find_closest_object_hit :360-406
    V3 p1t = t.p1 - ray.origin;
    V3 p2t = t.p2 - ray.origin;
    V3 p3t = t.p3 - ray.origin;
    p1t = permute(p1t, ray.kx, ray.ky, ray.kz);
    p2t = permute(p2t, ray.kx, ray.ky, ray.kz);
    p3t = permute(p3t, ray.kx, ray.ky, ray.kz);
    p1t.x += ray.Sx * p1t.z;
    p1t.y += ray.Sy * p1t.z;
    p2t.x += ray.Sx * p2t.z;
    p2t.y += ray.Sy * p2t.z;
    p3t.x += ray.Sx * p3t.z;
    p3t.y += ray.Sy * p3t.z;
    double e1 = p2t.x * p3t.y - p2t.y * p3t.x;
    double e2 = p3t.x * p1t.y - p3t.y * p1t.x;
    double e3 = p1t.x * p2t.y - p1t.y * p2t.x;
    if ((e1 < 0 || e2 < 0 || e3 < 0) && (e1 > 0 || e2 > 0 || e3 > 0)) return false;
    double det = e1 + e2 + e3;
    if (det == 0) return false;
    p1t.z *= ray.Sz;
    p2t.z *= ray.Sz;
    p3t.z *= ray.Sz;
    double t_scaled = e1 * p1t.z + e2 * p2t.z + e3 * p3t.z;
    if (det < 0 && (t_scaled >= 0 || t_scaled < ray.t_max * det)) return false;
    else if (det > 0 && (t_scaled <= 0 || t_scaled > ray.t_max * det)) return false;
    distance = t_scaled / det;
    return true;
}

// ---------------------------------------------------------------------------------
// BVH (src/bvh.cpp)
// ---------------------------------------------------------------------------------
V3 tri_max_point(const Tri& t) { // Triangle::max_axis_point :301-307
    return V3(std::max(std::max(t.p1.x, t.p2.x), t.p3.x), std::max(std::max(t.p1.y, t.p2.y), t.p3.y),
              std::max(std::max(t.p1.z, t.p2.z), t.p3.z));
}
V3 tri_min_point(const Tri& t) {
    return V3(std::min(std::min(t.p1.x, t.p2.x), t.p3.x), std::min(std::min(t.p1.y, t.p2.y), t.p3.y),
              std::min(std::min(t.p1.z, t.p2.z), t.p3.z));
}

V3 bounds_max(const std::vector<Tri>& tris, const int* ids, int n) { // get_max_point :5-17
    if (n == 0) return V3(0, 0, 0);
    V3 m = tri_max_point(tris[static_cast<size_t>(ids[0])]);
    for (int i = 1; i < n; i++) {
        V3 p = tri_max_point(tris[static_cast<size_t>(ids[i])]);
        m.x = std::max(p.x, m.x);
        m.y = std::max(p.y, m.y);
        m.z = std::max(p.z, m.z);
    }
    return m + V3(EPSILON);
}
V3 bounds_min(const std::vector<Tri>& tris, const int* ids, int n) { // get_min_point :19-31
    if (n == 0) return V3(0, 0, 0);
    V3 m = tri_min_point(tris[static_cast<size_t>(ids[0])]);
    for (int i = 1; i < n; i++) {
        V3 p = tri_min_point(tris[static_cast<size_t>(ids[i])]);
        m.x = std::min(p.x, m.x);
        m.y = std::min(p.y, m.y);
        m.z = std::min(p.z, m.z);
    }
    return m - V3(EPSILON);
}

int ceil_division(int a, int b) { return (a - 1) / b + 1; } // src/utils.cpp:44-46

int sah_split(const std::vector<Tri>& tris, const int* ids, int n, int n_axis_splits) { // surface_area_heuristic :91-127
    if (n <= 1) throw std::runtime_error("Cannot split triangle array of size 1 or less!");
    int split_index = 0;
    int bucket_width = ceil_division(n - (n_axis_splits - 1), n_axis_splits);
    double best_metric = -1;
    for (int i = bucket_width; i < n - bucket_width; i += bucket_width + 1) {
        V3 lmax = bounds_max(tris, ids, i + 1), lmin = bounds_min(tris, ids, i + 1);
        double lw = lmax.x - lmin.x, ld = lmax.y - lmin.y, lh = lmax.z - lmin.z;
        V3 rmax = bounds_max(tris, ids + i, n - (i + 1)), rmin = bounds_min(tris, ids + i, n - (i + 1));
        double rw = rmax.x - rmin.x, rd = rmax.y - rmin.y, rh = rmax.z - rmin.z;
        double LSA = 2 * lw * ld + 2 * lw * lh + 2 * ld * lh;
        double RSA = 2 * rw * rd + 2 * rw * rh + 2 * rd * rh;
        double metric = LSA * (i + 1) + RSA * (n - (i + 1));
        if (metric < best_metric || best_metric == -1) {
            best_metric = metric;
            split_index = i;
        }
    }
    return split_index;
}

// Node::Node :129-165.  `ids` is sorted in place, which for the root is the mesh's own
// primitive array (that is what fixes primitive_ID, src/objectunion.cpp:33-43).
int build_node(Mesh& mesh, const std::vector<Tri>& tris, int* ids, int n, int leaf_size, int n_axis_splits) {
    int index = static_cast<int>(mesh.nodes.size());
    mesh.nodes.emplace_back();
    {
        BvhNode& node = mesh.nodes[static_cast<size_t>(index)];
        node.lo = bounds_min(tris, ids, n);
        node.hi = bounds_max(tris, ids, n);
        if (n <= leaf_size) {
            node.leaf = true;
            node.prims.assign(ids, ids + n);
            return index;
        }
    }
    V3 lo = mesh.nodes[static_cast<size_t>(index)].lo, hi = mesh.nodes[static_cast<size_t>(index)].hi;
    double axis_length[3] = {hi.x - lo.x, hi.y - lo.y, hi.z - lo.z};
    int axis = 0; // get_split_axis :167-177
    double max_length = 0;
    for (int i = 0; i < 3; i++) {
        if (axis_length[i] >= max_length) {
            axis = i;
            max_length = axis_length[i];
        }
    }
    std::sort(ids, ids + n, [&tris, axis](int a, int b) { // sort_by_axis :85-89
        const Tri& ta = tris[static_cast<size_t>(a)];
        const Tri& tb = tris[static_cast<size_t>(b)];
        return ((ta.p1 + ta.p2 + ta.p3) / 3.0)[axis] < ((tb.p1 + tb.p2 + tb.p3) / 3.0)[axis];
    });
    int split_index = sah_split(tris, ids, n, n_axis_splits);
    int left_size = split_index + 1;
    int right_size = n - split_index - 1;
    std::vector<int> left(ids, ids + left_size), right(ids + left_size, ids + left_size + right_size);
    int c0 = build_node(mesh, tris, left.data(), left_size, leaf_size, n_axis_splits);
    int c1 = build_node(mesh, tris, right.data(), right_size, leaf_size, n_axis_splits);
    mesh.nodes[static_cast<size_t>(index)].child[0] = c0;
    mesh.nodes[static_cast<size_t>(index)].child[1] = c1;
    return index;
}

bool box_intersect(const BvhNode& node, const Ray& ray, double& distance) { // BoundingBox::intersect :60-83
    double imin = 0, imax = INF;
    for (int axis = 0; axis < 3; axis++) {
        const double d_inv = 1.0 / ray.dir[axis];
        const double t0 = (node.lo[axis] - ray.origin[axis]) * d_inv;
        const double t1 = (node.hi[axis] - ray.origin[axis]) * d_inv;
        const double t_min = std::min(t0, t1);
        const double t_max = std::max(t0, t1);
        imin = std::max(imin, t_min);
        imax = std::min(imax, t_max);
        if (imax <= imin) return false;
    }
    distance = std::max(imin, EPSILON);
    return true;
}

struct MeshHit {
    double distance = INF;
    int primitive = -1;
};

bool node_intersect(const Mesh& mesh, int index, Ray& ray, MeshHit& hit, Ctx& ctx) { // Node::intersect :179-222
    const BvhNode& node = mesh.nodes[static_cast<size_t>(index)];
    ctx.stats.bvh_nodes_visited++;
    if (node.leaf) {
        if (node.prims.empty()) return false;
        // find_closest_hit over the leaf's triangles (src/objects.cpp:414-442)
        double closest = INF;
        int closest_prim = -1;
        bool found = false;
        ray.prepare();
        for (int prim : node.prims) {
            double d;
            ctx.stats.triangles_tested++;
            bool success = tri_intersect(mesh.tris[static_cast<size_t>(prim)], ray, d);
            if (success && d > EPSILON && d < closest) {
                closest = d;
                closest_prim = prim;
                ray.t_max = d;
                found = true;
            }
        }
        if (found && closest < hit.distance && closest > EPSILON) {
            hit.distance = closest;
            hit.primitive = closest_prim;
            return true;
        }
        return false;
    }
    double d1, d2;
    bool hit1 = box_intersect(mesh.nodes[static_cast<size_t>(node.child[0])], ray, d1);
    bool hit2 = box_intersect(mesh.nodes[static_cast<size_t>(node.child[1])], ray, d2);
    bool s1 = false, s2 = false;
    if (hit1 && hit2) {
        if (d1 < d2) {
            s1 = node_intersect(mesh, node.child[0], ray, hit, ctx);
            if (d2 < hit.distance || hit.distance == -1) s2 = node_intersect(mesh, node.child[1], ray, hit, ctx);
        }
        else {
            s2 = node_intersect(mesh, node.child[1], ray, hit, ctx);
            if (d1 < hit.distance || hit.distance == -1) s1 = node_intersect(mesh, node.child[0], ray, hit, ctx);
        }
        return s1 || s2;
    }
    else if (hit1) {
        return node_intersect(mesh, node.child[0], ray, hit, ctx);
    }
    else if (hit2) {
        return node_intersect(mesh, node.child[1], ray, hit, ctx);
    }
    return false;
}

// ---------------------------------------------------------------------------------
// objects (src/objects.cpp)
// ---------------------------------------------------------------------------------
bool object_intersect(const OracleScene& s, const Object& o, Ray& ray, double& distance, int& primitive, Ctx& ctx) {
    switch (o.type) {
        case PTK_OBJ_SPHERE: { // Sphere::find_closest_object_hit :106-119
            double dot_product = dot(ray.dir, ray.origin);
            double b = 2 * (dot_product - dot(ray.dir, o.position));
            V3 difference = o.position - ray.origin;
            double c = len2(difference) - o.radius * o.radius;
            double d;
            bool success = solve_quadratic(b, c, d);
            if (!success || d > ray.t_max) return false;
            primitive = 0;
            distance = d;
            return true;
        }
        case PTK_OBJ_PLANE:
        case PTK_OBJ_RECTANGLE: { // Plane::compute_distance_in_centered_system :185-200
            V3 shifted = ray.origin - o.position;
            double direction_dot_normal = -dot(ray.dir, o.normal);
            if (std::abs(direction_dot_normal) < EPSILON) return false;
            double distances_to_start = dot(shifted, o.normal);
            double d = distances_to_start / direction_dot_normal;
            if (d < EPSILON) return false;
            if (d > ray.t_max) return false;
            if (o.type == PTK_OBJ_RECTANGLE) { // Rectangle::find_closest_object_hit :237-255
                double direction_dot_v1 = dot(ray.dir, o.v1);
                double direction_dot_v2 = dot(ray.dir, o.v2);
                double start_dot_v1 = dot(shifted, o.v1);
                double start_dot_v2 = dot(shifted, o.v2);
                if (std::abs(start_dot_v1 + direction_dot_v1 * d) > o.L1 / 2.0 + EPSILON ||
                    std::abs(start_dot_v2 + direction_dot_v2 * d) > o.L2 / 2.0 + EPSILON)
                    return false;
            }
            primitive = 0;
            distance = d;
            return true;
        }
        case PTK_OBJ_TRIANGLE: {
            double d;
            if (!tri_intersect(o.tri, ray, d)) return false;
            primitive = 0;
            distance = d;
            return true;
        }
        case PTK_OBJ_MESH: { // ObjectUnion::find_closest_object_hit :83-94 -> BVH::intersect src/bvh.cpp:229-237
            const Mesh& mesh = s.meshes[static_cast<size_t>(o.mesh)];
            double box_distance;
            if (mesh.nodes.empty() || !box_intersect(mesh.nodes[0], ray, box_distance)) return false;
            MeshHit mh;
            if (!node_intersect(mesh, 0, ray, mh, ctx)) return false;
            primitive = mh.primitive;
            distance = mh.distance;
            return true;
        }
    }
    return false;
}

const Tri& object_triangle(const OracleScene& s, const Object& o, int primitive) {
    if (o.type == PTK_OBJ_MESH) return s.meshes[static_cast<size_t>(o.mesh)].tris[static_cast<size_t>(primitive)];
    return o.tri;
}

V3 object_normal(const OracleScene& s, const Object& o, const V3& point, int primitive) { // get_normal_vector
    switch (o.type) {
        case PTK_OBJ_SPHERE: return normalize(point - o.position); // :121-124
        case PTK_OBJ_PLANE:
        case PTK_OBJ_RECTANGLE: return o.normal; // :213-215
        default: return tri_normal(object_triangle(s, o, primitive), point);
    }
}

V3 object_uv(const OracleScene& s, const Object& o, const V3& point, int primitive) { // get_UV
    switch (o.type) {
        case PTK_OBJ_SPHERE: { // :96-104
            V3 unit = (point - o.position) / o.radius;
            double x = -unit.x, y = -unit.y, z = -unit.z;
            double u = 0.5 + std::atan2(z, x) / (2 * M_PI);
            double v = 0.5 + std::asin(y) / (M_PI);
            return V3(u, v, 0);
        }
        case PTK_OBJ_PLANE: { // :178-183
            V3 shifted = point - o.position;
            return V3(1 - dot(shifted, o.v1) - 0.5, 1 - dot(shifted, o.v2) - 0.5, 0);
        }
        case PTK_OBJ_RECTANGLE: { // :230-235
            V3 shifted = point - o.position;
            return V3(1 - dot(shifted, o.v1) / o.L1 - 0.5, 1 - dot(shifted, o.v2) / o.L2 - 0.5, 0);
        }
        default: return tri_uv(object_triangle(s, o, primitive), point);
    }
}

bool find_closest_hit(Ctx& ctx, Hit& closest, Ray& ray) { // :414-442
    const OracleScene& s = *ctx.s;
    if (ctx.shadow) ctx.stats.shadow_rays++;
    else ctx.stats.extension_rays++;
    closest.distance = INF;
    bool found = false;
    ray.prepare();
    for (size_t i = 0; i < s.objects.size(); i++) {
        double d = INF;
        int primitive = -1;
        bool success = object_intersect(s, s.objects[i], ray, d, primitive, ctx);
        if (success && d > EPSILON && d < closest.distance) {
            closest.object = static_cast<int>(i);
            closest.primitive = primitive;
            closest.distance = d;
            ray.t_max = d;
            found = true;
        }
    }
    if (!found) return false;
    closest.point = ray.origin + ray.dir * closest.distance;
    V3 normal = object_normal(s, s.objects[static_cast<size_t>(closest.object)], closest.point, closest.primitive);
    closest.outside = dot(ray.dir, normal) < 0;
    closest.normal = closest.outside ? normal : -normal;
    closest.incident = ray.dir;
    return true;
}

// ---------------------------------------------------------------------------------
// materials (src/materials.cpp)
// ---------------------------------------------------------------------------------
struct Brdf { // BrdfData include/materials.h:13-18
    int type = DIFFUSE;
    double pdf = 0;
    V3 outgoing, brdf_over_pdf;
};

double medium_ior(const OracleScene& s, int medium) {
    return medium >= 0 ? s.media[static_cast<size_t>(medium)].ior : 1.0;
}

bool sample_transparency_map(const OracleScene& s, const Material& m, double u, double v, Rng& rng) { // :76-81
    double transparency = map1(s, m.transparency, u, v);
    double r = rng.uniform(0, 1);
    return r <= transparency;
}

bool material_allow_direct_light(const OracleScene& s, const Material& m, double u, double v, Rng& rng) {
    if (sample_transparency_map(s, m, u, v, rng)) return true; // :55-60
    if (m.type == PTK_MAT_TRANSPARENT) {                       // :129-136
        double n1 = medium_ior(s, m.internal_medium);
        double n2 = medium_ior(s, m.external_medium);
        return n1 == n2;
    }
    return false;
}

V3 material_emittance(const OracleScene& s, const Material& m, double u, double v) { // :72-74
    return map3(s, m.emission, u, v) * map1(s, m.intensity, u, v);
}

double get_alpha(const OracleScene& s, const Material& m, double u, double v) { // :185-188
    return std::max(map1(s, m.roughness, u, v), EPSILON);
}

double chi(double x) { return x > 0 ? 1 : 0; } // :181-183

double beckmann_D(const V3& half_vector, const V3& normal, double alpha) { // :190-200
    double cos_theta = std::fmin(dot(half_vector, normal), 1);
    double cos_theta_2 = cos_theta * cos_theta;
    double cos_theta_4 = cos_theta_2 * cos_theta_2;
    double tan_theta = std::sqrt((1.0 - cos_theta_2) / (cos_theta_2));
    double chi_factor = chi(cos_theta);
    double alpha_b_2 = alpha * alpha;
    double fraction = 1.0 / (M_PI * alpha_b_2 * cos_theta_4);
    double exp_factor = std::exp(-(tan_theta * tan_theta) / alpha_b_2);
    return chi_factor * fraction * exp_factor;
}

double smith_G1(const V3& half_vector, const V3& normal, const V3& v, double alpha) { // :202-212
    double cos_theta = dot(half_vector, v);
    double tan_theta = std::sqrt((1.0 - cos_theta * cos_theta) / (cos_theta * cos_theta));
    double a = 1.0 / (alpha * tan_theta);
    double approximation = a < 1.6 ? (3.535 * a + 2.181 * a * a) / (1.0 + 2.276 * a + 2.577 * a * a) : 1.0;
    return chi(cos_theta / dot(v, normal)) * approximation;
}

double smith_G(const V3& h, const V3& n, const V3& incident, const V3& outgoing, double alpha) { // :214-218
    return smith_G1(h, n, -incident, alpha) * smith_G1(h, n, outgoing, alpha);
}

V3 sample_half_vector(const V3& normal, double alpha, Rng& rng) { // :220-235
    double r1 = rng.uniform(0, 1);
    double r2 = rng.uniform(0, 1);
    double phi = 2 * M_PI * r2;
    double tan_theta2 = -alpha * alpha * std::log(1 - r1);
    double cos_theta2 = 1.0 / (1.0 + tan_theta2);
    double cos_theta = std::sqrt(cos_theta2);
    double sin_theta = std::sqrt(1 - cos_theta2);
    V3 x_hat, y_hat;
    perpendicular_vectors(normal, x_hat, y_hat);
    return x_hat * sin_theta * std::cos(phi) + y_hat * sin_theta * std::sin(phi) + normal * cos_theta;
}

double diffuse_pdf(const V3& outgoing, const V3& normal) { // :237-239
    return std::max(dot(normal, outgoing) / M_PI, 0.0);
}

double specular_pdf(const OracleScene& s, const Material& m, const V3& outgoing, const V3& incident, const V3& normal,
                    double u, double v) { // :241-247
    V3 h = normalize(outgoing - incident);
    double half_vector_pdf = beckmann_D(h, normal, get_alpha(s, m, u, v)) * dot(h, normal);
    return std::max(half_vector_pdf / (4.0 * dot(outgoing, h)), 0.0);
}

double material_pdf(const OracleScene& s, const Material& m, const V3& outgoing, const V3& incident, const V3& normal,
                    double u, double v) { // brdf_pdf
    switch (m.type) {
        case PTK_MAT_DIFFUSE: return std::max(dot(outgoing, normal), 0.0) / M_PI; // :103-106
        case PTK_MAT_GLOSSY:                                                      // :316-320
            return 0.5 * (diffuse_pdf(outgoing, normal) + specular_pdf(s, m, outgoing, incident, normal, u, v));
        case PTK_MAT_METALLIC_MICROFACET:
        case PTK_MAT_REFLECTIVE_MICROFACET: return specular_pdf(s, m, outgoing, incident, normal, u, v); // :385-388
        default: return 0; // Reflective :125-127, Transparent (base :68-70), TransparentMicrofacet :457-460
    }
}

V3 material_eval(const OracleScene& s, const Material& m, const Hit& hit, const V3& outgoing, double u, double v,
                 Rng& rng) {
    switch (m.type) {
        case PTK_MAT_DIFFUSE: // :83-88
            if (sample_transparency_map(s, m, u, v, rng)) return V3(0, 0, 0);
            return map3(s, m.albedo, u, v) / M_PI;
        case PTK_MAT_GLOSSY: { // :249-286
            if (sample_transparency_map(s, m, u, v, rng)) return V3(0, 0, 0);
            V3 h = normalize(outgoing - hit.incident);
            double n1, n2;
            if (hit.outside) {
                n1 = medium_ior(s, m.external_medium);
                n2 = m.n;
            }
            else {
                n1 = m.n;
                n2 = medium_ior(s, m.internal_medium);
            }
            double i_dot_h = -dot(hit.incident, h);
            double F_r = schlick_fresnel(i_dot_h, n1, n2);
            double R_0_sqrt = (n1 - n2) / (n1 + n2);
            double R_0 = R_0_sqrt * R_0_sqrt;
            double fac1 = std::min((1.0 - dot(hit.normal, -hit.incident) / 2.0), 1.0);
            double fac2 = std::min((1.0 - dot(hit.normal, outgoing) / 2.0), 1.0);
            V3 diffuse = map3(s, m.albedo, u, v) * 28.0 / (23.0 * M_PI) * (1 - R_0) *
                         (1 - fac1 * fac1 * fac1 * fac1 * fac1) * (1 - fac2 * fac2 * fac2 * fac2 * fac2);
            double alpha = get_alpha(s, m, u, v);
            V3 reflection_color = m.is_dielectric ? V3(1.0, 1.0, 1.0) : map3(s, m.albedo, u, v);
            double d_factor = beckmann_D(h, hit.normal, alpha) * dot(h, hit.normal);
            double g_factor = smith_G(h, hit.normal, hit.incident, outgoing, alpha);
            double denom_factor = -1.0 / (4.0 * dot(hit.incident, hit.normal) * dot(hit.normal, outgoing));
            V3 specular = reflection_color * F_r * d_factor * g_factor * denom_factor;
            return diffuse + specular;
        }
        case PTK_MAT_METALLIC_MICROFACET: { // :329-361 (no transparency test)
            V3 h = normalize(outgoing - hit.incident);
            double n1, k1, n2, k2;
            if (hit.outside) {
                n1 = medium_ior(s, m.external_medium);
                k1 = 0;
                n2 = m.n;
                k2 = m.k;
            }
            else {
                n1 = m.n;
                k1 = m.k;
                n2 = medium_ior(s, m.external_medium);
                k2 = 0;
            }
            double i_dot_h = -dot(hit.incident, h);
            double F_r = fresnel_multiplier(i_dot_h, n1, k1, n2, k2, false);
            double alpha = get_alpha(s, m, u, v);
            V3 reflection_color = map3(s, m.albedo, u, v);
            double d_factor = beckmann_D(h, hit.normal, alpha) * dot(h, hit.normal);
            double g_factor = smith_G(h, hit.normal, hit.incident, outgoing, alpha);
            double denom_factor = -1.0 / (4.0 * dot(hit.incident, hit.normal) * dot(hit.normal, outgoing));
            return reflection_color * F_r * d_factor * g_factor * denom_factor;
        }
        case PTK_MAT_REFLECTIVE_MICROFACET: { // :390-404
            if (sample_transparency_map(s, m, u, v, rng)) return V3(0, 0, 0);
            V3 h = normalize(outgoing - hit.incident);
            double alpha = get_alpha(s, m, u, v);
            double d_factor = beckmann_D(h, hit.normal, alpha) * dot(h, hit.normal);
            double g_factor = smith_G(h, hit.normal, hit.incident, outgoing, alpha);
            double denom_factor = -1.0 / (4.0 * dot(hit.incident, hit.normal) * dot(hit.normal, outgoing));
            return map3(s, m.albedo, u, v) * d_factor * g_factor * denom_factor;
        }
        default: return V3(0, 0, 0); // Reflective :108-110, Transparent :138-140, TransparentMicrofacet :406-408
    }
}

Brdf transparent_ray(const Hit& hit) { // sample_transparent_ray :45-52
    Brdf data;
    data.type = TRANSMITTED;
    data.outgoing = hit.incident;
    data.brdf_over_pdf = V3(1.0, 1.0, 1.0);
    data.pdf = 0;
    return data;
}

Brdf material_sample(const OracleScene& s, const Material& m, const Hit& hit, double u, double v, Rng& rng) {
    if (sample_transparency_map(s, m, u, v, rng)) return transparent_ray(hit);
    Brdf data;
    switch (m.type) {
        case PTK_MAT_DIFFUSE: { // :90-101
            V3 outgoing = sample_cosine_hemisphere(hit.normal, rng);
            data.outgoing = outgoing;
            data.brdf_over_pdf = map3(s, m.albedo, u, v);
            data.pdf = material_pdf(s, m, outgoing, hit.incident, hit.normal, u, v);
            data.type = DIFFUSE;
            return data;
        }
        case PTK_MAT_REFLECTIVE: { // :112-123
            data.outgoing = reflect_vector(hit.incident, hit.normal);
            data.brdf_over_pdf = m.is_dielectric ? V3(1.0, 1.0, 1.0) : map3(s, m.albedo, u, v);
            data.pdf = 0;
            data.type = REFLECTED;
            return data;
        }
        case PTK_MAT_TRANSPARENT: { // :142-179
            double n1, n2;
            if (hit.outside) {
                n1 = medium_ior(s, m.external_medium);
                n2 = medium_ior(s, m.internal_medium);
            }
            else {
                n1 = medium_ior(s, m.internal_medium);
                n2 = medium_ior(s, m.external_medium);
            }
            V3 transmitted = refract_vector(hit.incident, -hit.normal, n1 / n2);
            double F_r = 1;
            if (len2(transmitted) != 0) {
                double cos_incident = -dot(hit.incident, hit.normal);
                F_r = fresnel_multiplier(cos_incident, n1, 0, n2, 0, true);
            }
            double random_num = rng.uniform(0, 1);
            bool is_reflected = random_num <= F_r;
            if (is_reflected) {
                data.type = REFLECTED;
                data.outgoing = reflect_vector(hit.incident, hit.normal);
            }
            else {
                data.type = TRANSMITTED;
                data.outgoing = transmitted;
            }
            data.brdf_over_pdf = V3(1.0, 1.0, 1.0);
            data.pdf = 0;
            return data;
        }
        case PTK_MAT_GLOSSY:                  // :288-314
        case PTK_MAT_METALLIC_MICROFACET:     // :363-383
        case PTK_MAT_REFLECTIVE_MICROFACET: { // inherits Metallic's sample
            if (m.type == PTK_MAT_GLOSSY) {
                double rand_1 = rng.uniform(0, 1);
                if (rand_1 <= 0.5) {
                    data.outgoing = sample_cosine_hemisphere(hit.normal, rng);
                }
                else {
                    V3 h = sample_half_vector(hit.normal, get_alpha(s, m, u, v), rng);
                    data.outgoing = reflect_vector(hit.incident, h);
                }
            }
            else {
                V3 h = sample_half_vector(hit.normal, get_alpha(s, m, u, v), rng);
                data.outgoing = reflect_vector(hit.incident, h);
            }
            data.pdf = material_pdf(s, m, data.outgoing, hit.incident, hit.normal, u, v);
            if (data.pdf == 0) data.brdf_over_pdf = V3(0.0);
            else
                data.brdf_over_pdf =
                    material_eval(s, m, hit, data.outgoing, u, v, rng) * dot(data.outgoing, hit.normal) / data.pdf;
            data.type = DIFFUSE;
            return data;
        }
        case PTK_MAT_TRANSPARENT_MICROFACET: { // :410-455
            double n1, n2;
            if (hit.outside) {
                n1 = medium_ior(s, m.external_medium);
                n2 = medium_ior(s, m.internal_medium);
            }
            else {
                n1 = medium_ior(s, m.internal_medium);
                n2 = medium_ior(s, m.external_medium);
            }
            V3 h = sample_half_vector(hit.normal, get_alpha(s, m, u, v), rng);
            double i_dot_h = -dot(hit.incident, h);
            double F_r = fresnel_multiplier(i_dot_h, n1, 0, n2, 0, true);
            V3 refracted = refract_vector(hit.incident, -h, n1 / n2);
            double rand_num = rng.uniform(0, 1);
            if (rand_num <= F_r || len2(refracted) == 0) data.outgoing = reflect_vector(hit.incident, h);
            else data.outgoing = refracted;
            data.type = TRANSMITTED;
            double cosine_factor = dot(hit.incident, h) / (dot(hit.incident, hit.normal) * dot(h, hit.normal));
            data.brdf_over_pdf =
                V3(smith_G(h, hit.normal, hit.incident, data.outgoing, get_alpha(s, m, u, v)) * cosine_factor);
            data.pdf = 0;
            return data;
        }
    }
    return data;
}

// ---------------------------------------------------------------------------------
// media (src/medium.cpp)
// ---------------------------------------------------------------------------------
double medium_sample_distance(const Medium& m, Rng& rng) {
    if (m.type != PTK_MEDIUM_HOMOGENEOUS_SCATTERING) return INF; // :13-15
    int channel = rng.integer(0, 3);                              // :45-51
    if (m.extinction[channel] == 0) return INF;
    return -std::log(rng.uniform(0, 1)) / m.extinction[channel];
}

V3 medium_transmittance(const Medium& m, double distance) { // :25-27
    return vexp(-m.extinction * distance);
}

V3 medium_sample(const Medium& m, double distance, bool scatter) {
    if (m.type == PTK_MEDIUM_BEERS_LAW) return medium_transmittance(m, distance); // :41-43
    V3 tr = medium_transmittance(m, distance);                                     // :53-64
    V3 density = scatter ? m.extinction * tr : tr;
    double pdf = 0;
    for (int i = 0; i < 3; i++) pdf += density[i];
    pdf *= 1.0 / 3.0;
    return scatter ? tr * m.scattering / pdf : tr / pdf;
}

V3 medium_emission(const Medium& m) {
    if (m.type != PTK_MEDIUM_HOMOGENEOUS_SCATTERING) return V3(0, 0, 0); // :33-35
    double pdf = 0;                                                        // :66-73
    for (int i = 0; i < 3; i++) pdf += m.extinction[i];
    pdf *= 1.0 / 3.0;
    return m.emission * m.absorption / pdf;
}

const double PHASE_ISOTROPIC = 1.0 / (4 * M_PI); // :21-23

// ---------------------------------------------------------------------------------
// light sampling (src/objects.cpp:68-166,257-265,408-412; src/objectunion.cpp:100-142)
// ---------------------------------------------------------------------------------
double area_to_angle_factor(const OracleScene& s, const Object& o, const V3& surface_point, const V3& from,
                            int primitive) { // :68-76
    V3 normal = object_normal(s, o, surface_point, primitive);
    V3 difference = from - surface_point;
    V3 to_point = normalize(difference);
    double pdf = dot(normal, to_point) / len2(difference);
    return std::abs(pdf);
}

double object_light_pdf(const OracleScene& s, const Object& o, const V3& surface_point, const V3& from, int primitive) {
    switch (o.type) {
        case PTK_OBJ_SPHERE: { // :130-138
            double distance = len(from - o.position);
            if (distance <= o.radius) return 1.0 / (o.area * area_to_angle_factor(s, o, surface_point, from, 0));
            double cos_theta_max = std::sqrt(1 - std::pow(o.radius / distance, 2));
            return 1.0 / (2.0 * M_PI * (1 - (cos_theta_max)));
        }
        case PTK_OBJ_PLANE: return 0.0; // :217-219
        case PTK_OBJ_RECTANGLE:         // :263-265
            return std::abs(1.0 / (o.area * area_to_angle_factor(s, o, surface_point, from, primitive)));
        case PTK_OBJ_TRIANGLE: // Object::light_pdf :78-80
            return 1.0 / (o.area * area_to_angle_factor(s, o, surface_point, from, primitive));
        case PTK_OBJ_MESH: { // ObjectUnion::light_pdf src/objectunion.cpp:132-135
            const Mesh& mesh = s.meshes[static_cast<size_t>(o.mesh)];
            return 1.0 / (mesh.cumulative_area[mesh.n_lights - 1] * area_to_angle_factor(s, o, surface_point, from, primitive));
        }
    }
    return 0;
}

V3 triangle_random_point(const Tri& t, Rng& rng) { // :408-412
    double r1 = rng.uniform(0, 1);
    double r2 = rng.uniform(0, 1);
    return t.p1 * (1.0 - std::sqrt(r1)) + t.p2 * (std::sqrt(r1) * (1.0 - r2)) + t.p3 * (std::sqrt(r1) * r2);
}

int mesh_random_primitive(const Mesh& mesh, Rng& rng) { // sample_random_primitive_index src/objectunion.cpp:100-126
    double random_area_split = rng.uniform(0, mesh.area);
    int max = static_cast<int>(mesh.n_lights) - 1;
    int min = 0;
    int index = min;
    if (mesh.cumulative_area[static_cast<size_t>(index)] >= random_area_split) return mesh.light_index[static_cast<size_t>(index)];
    while (min <= max) {
        index = (max + min) / 2;
        if (mesh.cumulative_area[static_cast<size_t>(index)] < random_area_split) {
            min = index + 1;
        }
        else if (mesh.cumulative_area[static_cast<size_t>(index)] == random_area_split ||
                 (mesh.cumulative_area[static_cast<size_t>(index)] >= random_area_split &&
                  mesh.cumulative_area[static_cast<size_t>(index - 1)] < random_area_split)) {
            break;
        }
        else {
            max = index - 1;
        }
    }
    return mesh.light_index[static_cast<size_t>(index)];
}

V3 object_random_light_point(const OracleScene& s, const Object& o, const V3& from, double& pdf, Rng& rng) {
    switch (o.type) {
        case PTK_OBJ_SPHERE: { // :140-166
            double distance = len(from - o.position);
            if (distance <= o.radius) {
                V3 random_point = sample_spherical(rng) * o.radius + o.position;
                pdf = object_light_pdf(s, o, random_point, from, 0);
                return random_point;
            }
            double cos_theta_max = std::sqrt(1 - std::pow(o.radius / distance, 2));
            pdf = object_light_pdf(s, o, V3(0.0), from, 0);
            double r = rng.uniform(0, 1);
            double cos_theta = 1 + r * (cos_theta_max - 1);
            double sin_theta = std::sqrt(1 - cos_theta * cos_theta);
            double cos_alpha =
                (o.radius * o.radius + distance * distance -
                 std::pow(distance * cos_theta - std::sqrt(o.radius * o.radius - std::pow(distance * sin_theta, 2)), 2)) /
                (2.0 * distance * o.radius);
            double sin_alpha = std::sqrt(1.0 - cos_alpha * cos_alpha);
            V3 x_hat, y_hat;
            V3 z_hat = object_normal(s, o, from, 0);
            perpendicular_vectors(z_hat, x_hat, y_hat);
            double phi = rng.uniform(0, 2.0 * M_PI);
            V3 random_point = x_hat * sin_alpha * std::cos(phi) + y_hat * sin_alpha * std::sin(phi) + z_hat * cos_alpha;
            return random_point * o.radius + o.position;
        }
        case PTK_OBJ_PLANE: { // Object::random_light_point :82-86 with the base surface point :64-66
            V3 random_point;
            pdf = object_light_pdf(s, o, random_point, from, 0);
            return random_point;
        }
        case PTK_OBJ_RECTANGLE: { // :257-261
            double r1 = rng.uniform(-o.L1 / 2, o.L1 / 2);
            double r2 = rng.uniform(-o.L2 / 2, o.L2 / 2);
            V3 random_point = o.v1 * r1 + o.v2 * r2 + o.position;
            pdf = object_light_pdf(s, o, random_point, from, 0);
            return random_point;
        }
        case PTK_OBJ_TRIANGLE: {
            V3 random_point = triangle_random_point(o.tri, rng);
            pdf = object_light_pdf(s, o, random_point, from, 0);
            return random_point;
        }
        case PTK_OBJ_MESH: { // ObjectUnion::random_light_point src/objectunion.cpp:137-142
            const Mesh& mesh = s.meshes[static_cast<size_t>(o.mesh)];
            int random_index = mesh_random_primitive(mesh, rng);
            V3 random_point = triangle_random_point(mesh.tris[static_cast<size_t>(random_index)], rng);
            pdf = object_light_pdf(s, o, random_point, from, random_index);
            return random_point;
        }
    }
    pdf = 0;
    return V3();
}

bool object_is_light(const OracleScene& s, const Object& o) { // is_light_source :27-29, src/objectunion.cpp:59-61
    return s.materials[static_cast<size_t>(o.material)].is_light;
}

double mis_weight(int n_a, double pdf_a, int n_b, double pdf_b) { // :464-468
    double f = n_a * pdf_a;
    double g = n_b * pdf_b;
    return f / (f + g);
}

V3 compute_visibility(Ctx& ctx, const V3& point, int current_medium, int light_index, const V3& direction,
                      V3& transmittance, double& distance) { // :470-513
    const OracleScene& s = *ctx.s;
    Ray ray;
    ray.origin = point;
    ray.dir = direction;
    transmittance = V3(1.0);
    V3 light_emittance(0.0);
    int medium = current_medium;
    distance = 0;
    ctx.shadow = true;
    while (true) {
        ray.t_max = INF;
        Hit light_hit;
        if (!find_closest_hit(ctx, light_hit, ray)) {
            ctx.shadow = false;
            return V3(0.0);
        }
        distance += light_hit.distance;
        if (medium < 0) medium = s.background_medium;
        transmittance *= medium_transmittance(s.media[static_cast<size_t>(medium)], light_hit.distance);
        const Object& o = s.objects[static_cast<size_t>(light_hit.object)];
        const Material& m = s.materials[static_cast<size_t>(o.material)];
        V3 uv = object_uv(s, o, light_hit.point, light_hit.primitive);
        if (light_hit.object == light_index) {
            light_emittance = material_emittance(s, m, uv.x, uv.y);
            break;
        }
        else if (!material_allow_direct_light(s, m, uv.x, uv.y, ctx.rng)) {
            ctx.shadow = false;
            return V3(0.0);
        }
        ray.origin = light_hit.point;
        bool leaving_object = !light_hit.outside;
        medium = leaving_object ? m.external_medium : m.internal_medium;
    }
    ctx.shadow = false;
    return light_emittance;
}

V3 sample_light(Ctx& ctx, const Hit& hit, int current_medium, bool is_scatter) { // :515-580
    const OracleScene& s = *ctx.s;
    V3 L(0.0);
    // sample_random_light :444-462
    int number_of_light_sources = 0;
    std::vector<int> lights;
    for (size_t i = 0; i < s.objects.size(); i++)
        if (object_is_light(s, s.objects[i])) lights.push_back(static_cast<int>(i));
    number_of_light_sources = static_cast<int>(lights.size());
    if (number_of_light_sources == 0) return L;
    int light_index = lights[static_cast<size_t>(ctx.rng.integer(0, number_of_light_sources))];
    if (light_index == hit.object) return L;

    const Object& light = s.objects[static_cast<size_t>(light_index)];
    double light_pdf;
    V3 random_point = object_random_light_point(s, light, hit.point, light_pdf, ctx.rng);
    if (light_pdf == 0) return L;

    V3 sampled_direction = random_point - hit.point;
    double distance_to_light = len(sampled_direction);
    sampled_direction = normalize(sampled_direction);

    V3 brdf;
    double scatter_pdf;
    if (!is_scatter) {
        const Object& o = s.objects[static_cast<size_t>(hit.object)];
        const Material& m = s.materials[static_cast<size_t>(o.material)];
        V3 uv = object_uv(s, o, hit.point, hit.primitive);
        brdf = material_eval(s, m, hit, sampled_direction, uv.x, uv.y, ctx.rng);
        if (len2(brdf) == 0) return L;
        V3 uv2 = object_uv(s, o, hit.point, hit.primitive);
        scatter_pdf = material_pdf(s, m, sampled_direction, hit.incident, hit.normal, uv2.x, uv2.y);
    }
    else {
        scatter_pdf = PHASE_ISOTROPIC;
    }

    double distance;
    V3 transmittance;
    ctx.rng.fork_begin();
    V3 emittance =
        compute_visibility(ctx, hit.point, current_medium, light_index, sampled_direction, transmittance, distance);
    ctx.rng.fork_end();

    if (std::abs(distance_to_light - distance) > EPSILON || len2(emittance) == 0) return L;

    double weight = mis_weight(1, light_pdf, 1, scatter_pdf);
    if (is_scatter) {
        L = weight * scatter_pdf * emittance * transmittance / light_pdf;
    }
    else {
        bool wrong_side = (dot(hit.incident, hit.normal) * dot(sampled_direction, hit.normal)) > 0;
        if (wrong_side) return L;
        double cosine = std::max(dot(hit.normal, sampled_direction), 0.0);
        L = weight * brdf * cosine * emittance * transmittance / light_pdf;
    }
    L *= static_cast<double>(number_of_light_sources);
    return L;
}

// ---------------------------------------------------------------------------------
// camera (src/camera.cpp)
// ---------------------------------------------------------------------------------
Ray make_ray(const OracleScene& s, double x, double y, Rng& rng) { // :24-70
    double W = static_cast<double>(s.settings.width), H = static_cast<double>(s.settings.height);
    double local_x = x * s.cam_width / W - s.cam_width / 2.0;
    V3 lx = s.cam_screen_x * local_x;
    double local_y = y * s.cam_screen_height / H - s.cam_screen_height / 2.0;
    V3 ly = s.cam_screen_y * local_y;
    V3 pixel_vector = s.cam_position + lx + ly + s.cam_view * s.cam_focal;
    V3 direction = normalize(pixel_vector - s.cam_position);
    Ray ray;
    ray.origin = s.cam_position;
    ray.dir = direction;
    if (s.cam_dof != PTK_DOF_NONE) { // adjust_depth_of_field :34-55
        double t = s.cam_focus / dot(ray.dir, s.cam_view);
        V3 focal_point = ray.origin + t * ray.dir;
        V3 adjust;
        if (s.cam_dof == PTK_DOF_CIRCLE) {
            double r = s.cam_aperture / 2.0 * std::sqrt(rng.uniform(0, 1));
            double phi = rng.uniform(0, 2 * M_PI);
            adjust = r * (std::cos(phi) * s.cam_screen_x + std::sin(phi) * s.cam_screen_y);
        }
        else {
            double r1 = rng.uniform(0, s.cam_aperture) - s.cam_aperture / 2.0;
            double r2 = rng.uniform(0, s.cam_aperture) - s.cam_aperture / 2.0;
            adjust = (r1 * s.cam_screen_y + r2 * s.cam_screen_x);
        }
        ray.origin += adjust;
        ray.dir = normalize(focal_point - ray.origin);
    }
    ray.type = TRANSMITTED;
    return ray;
}

Ray camera_sample(const OracleScene& s, size_t x, size_t y, Rng& rng) { // compute_pixel_color app/pathtracer.cpp:182-189
    double new_x = static_cast<double>(x);
    double new_y = static_cast<double>(y);
    if (s.settings.enable_anti_aliasing) {
        new_x += rng.normal() / 3.0;
        new_y += rng.normal() / 3.0;
    }
    return make_ray(s, new_x, new_y, rng);
}

// ---------------------------------------------------------------------------------
// integrator (app/pathtracer.cpp:39-176)
// ---------------------------------------------------------------------------------
struct PixelData {
    V3 color, position, normal;
};

PixelData raytrace(Ctx& ctx, Ray ray) {
    const OracleScene& s = *ctx.s;
    PixelData data;
    V3 color(0, 0, 0);
    V3 throughput(1, 1, 1);
    double random_threshold = 1;
    bool allow_recursion = true;
    bool has_hit_surface = false;
    V3 saved_point;
    double scatter_pdf = 0;
    int medium = s.background_medium;
    const bool nee = s.settings.enable_next_event_estimation != 0;
    ctx.stats.paths++;

    for (int depth = 0; depth <= s.settings.max_recursion_depth; depth++) {
        if (medium < 0) medium = s.background_medium;
        const Medium& med = s.media[static_cast<size_t>(medium)];
        double scatter_distance = medium_sample_distance(med, ctx.rng);
        ray.t_max = scatter_distance;
        Hit ray_hit;
        if (!find_closest_hit(ctx, ray_hit, ray)) {
            if (scatter_distance == INF) {
                color += throughput * s.background_color;
                break;
            }
            ray_hit.distance = INF;
        }
        bool scatter = scatter_distance < ray_hit.distance;
        scatter_distance = std::min(scatter_distance, ray_hit.distance);
        if (scatter) color += medium_emission(med) * throughput;
        throughput *= medium_sample(med, scatter_distance, scatter);

        if (scatter) {
            ctx.stats.scatter_vertices++;
            V3 scatter_point = ray.origin + ray.dir * scatter_distance;
            V3 scattered_direction = sample_spherical(ctx.rng);
            if (nee) {
                ray_hit.point = scatter_point;
                color += sample_light(ctx, ray_hit, medium, true) * throughput;
                ray.type = DIFFUSE;
                scatter_pdf = PHASE_ISOTROPIC;
                saved_point = scatter_point;
            }
            ray.origin = scatter_point;
            ray.dir = scattered_direction;
        }
        else {
            ctx.stats.surface_vertices++;
            if (!has_hit_surface) {
                data.position = ray_hit.point;
                data.normal = ray_hit.normal;
                has_hit_surface = true;
            }
            bool is_specular_ray = ray.type == REFLECTED || ray.type == TRANSMITTED;
            const Object& hit_object = s.objects[static_cast<size_t>(ray_hit.object)];
            const Material& material = s.materials[static_cast<size_t>(hit_object.material)];
            ctx.stats.vertices_by_material[material.type]++;

            if (material.is_light) {
                double weight;
                if (!nee || depth == 0 || is_specular_ray) {
                    weight = 1;
                }
                else {
                    double light_pdf = object_light_pdf(s, hit_object, ray_hit.point, saved_point, ray_hit.primitive);
                    weight = mis_weight(1, scatter_pdf, 1, light_pdf);
                }
                V3 uv = object_uv(s, hit_object, ray_hit.point, ray_hit.primitive);
                V3 light_emittance = material_emittance(s, material, uv.x, uv.y);
                color += weight * light_emittance * throughput;
            }

            if (nee) color += sample_light(ctx, ray_hit, medium, false) * throughput;

            V3 uv = object_uv(s, hit_object, ray_hit.point, ray_hit.primitive);
            Brdf brdf_result = material_sample(s, material, ray_hit, uv.x, uv.y, ctx.rng);
            bool is_virtual_surface = material_allow_direct_light(s, material, uv.x, uv.y, ctx.rng);
            if (is_virtual_surface) {
                brdf_result.type = ray.type;
            }
            else {
                scatter_pdf = brdf_result.pdf;
                saved_point = ray_hit.point;
            }
            throughput *= brdf_result.brdf_over_pdf;

            double incoming_dot_normal = dot(ray_hit.incident, ray_hit.normal);
            double outgoing_dot_normal = dot(brdf_result.outgoing, ray_hit.normal);
            bool penetrating_boundary = incoming_dot_normal * outgoing_dot_normal > 0;
            if (penetrating_boundary) {
                medium = ray_hit.outside ? material.internal_medium : material.external_medium;
            }
            ray.origin = ray_hit.point;
            ray.dir = brdf_result.outgoing;
            ray.type = brdf_result.type;
        }

        if (depth < s.settings.min_recursion_steps) {
            random_threshold = 1;
            allow_recursion = true;
        }
        else {
            random_threshold = std::min(vmax(throughput), 0.9);
            double random_value = ctx.rng.uniform(0, 1);
            allow_recursion = random_value < random_threshold;
        }
        if (!allow_recursion) break;
        throughput /= random_threshold;
    }
    data.color = color;
    return data;
}

double increment_average(double previous, double addition, double samples_prev, double samples_add) { // :232-235
    return (previous * samples_prev + addition * samples_add) / (samples_prev + samples_add);
}

void init_rng(Rng& rng, const OracleRng* cfg) {
    rng.mode = cfg ? cfg->mode : ORACLE_RNG_COUNTER;
    if (rng.mode == ORACLE_RNG_REFERENCE_STREAMS) {
        rng.ugen.seed(cfg->uniform_seed);
        rng.ngen.seed(cfg->normal_seed);
        rng.udist.reset();
        rng.ndist.reset();
        srand(cfg->rand_seed);
    }
}

void merge_stats(OracleScene* scene, const OracleStats& st) {
    std::lock_guard<std::mutex> lock(scene->stats_mutex);
    OracleStats& t = scene->stats;
    t.paths += st.paths;
    t.extension_rays += st.extension_rays;
    t.shadow_rays += st.shadow_rays;
    t.surface_vertices += st.surface_vertices;
    t.scatter_vertices += st.scatter_vertices;
    t.bvh_nodes_visited += st.bvh_nodes_visited;
    t.triangles_tested += st.triangles_tested;
    for (int i = 0; i < PTK_MAT_COUNT; i++) t.vertices_by_material[i] += st.vertices_by_material[i];
}

int set_error(int code, const std::string& message) {
    g_error = message;
    return code;
}

bool check_job(const OracleScene* scene, const PtkRenderJob* job) {
    if (!scene || !job) return false;
    int64_t n = static_cast<int64_t>(scene->settings.width) * scene->settings.height;
    return job->start_idx >= 0 && job->number_of_pixels >= 0 && job->start_idx + job->number_of_pixels <= n &&
           job->number_of_samples >= 0 && job->sample_offset >= 0;
}

// one pixel of raytrace_section (app/pathtracer.cpp:239-268) without the accumulation
void pixel_coordinates(const OracleScene& s, int64_t idx, size_t& x, size_t& y) {
    size_t W = static_cast<size_t>(s.settings.width), H = static_cast<size_t>(s.settings.height);
    x = static_cast<size_t>(idx) % W;
    y = H - static_cast<size_t>(idx) / W;
}

} // namespace

// -------------------------------------------------------------------------------------
// C entry points
// -------------------------------------------------------------------------------------
extern "C" {

const char* oracle_last_error(void) {
    return g_error.c_str();
}

int oracle_scene_create(const PtkSceneDesc* d, OracleScene** out) {
    if (!d || !out) return set_error(PTK_ERR_INVALID, "oracle_scene_create: null argument");
    try {
        std::unique_ptr<OracleScene> s(new OracleScene());
        s->settings = d->settings;
        s->map_data.assign(d->map_data, d->map_data + d->n_map_values);
        for (int i = 0; i < d->n_maps; i++) {
            Map m;
            m.width = d->maps[i].width;
            m.height = d->maps[i].height;
            m.dim = d->maps[i].dim;
            m.data = s->map_data.data() + d->maps[i].offset;
            s->maps.push_back(m);
        }
        for (int i = 0; i < d->n_media; i++) { // Medium::Medium src/medium.cpp:4-11, BeersLawMedium :37-39
            const PtkMedium& pm = d->media[i];
            Medium m;
            m.type = pm.type;
            m.ior = pm.refractive_index;
            m.absorption = V3(pm.absorption);
            m.scattering = pm.type == PTK_MEDIUM_BEERS_LAW ? V3(0.0) : V3(pm.scattering);
            m.emission = V3(pm.emission);
            m.extinction = m.absorption + m.scattering;
            s->media.push_back(m);
        }
        for (int i = 0; i < d->n_materials; i++) { // Material::Material src/materials.cpp:3-43
            const PtkMaterial& pm = d->materials[i];
            Material m;
            m.type = pm.type;
            m.is_light = pm.is_light_source != 0;
            m.is_dielectric = pm.is_dielectric != 0;
            if (pm.type == PTK_MAT_METALLIC_MICROFACET || pm.type == PTK_MAT_REFLECTIVE_MICROFACET)
                m.is_dielectric = false; // ensure_metallic :322-327
            m.n = pm.surface_refractive_index;
            m.k = m.is_dielectric ? 0 : pm.extinction_coefficient;
            m.albedo = pm.albedo_map;
            m.emission = pm.emission_color_map;
            m.intensity = pm.light_intensity_map;
            m.roughness = pm.roughness_map;
            m.transparency = pm.transparency_map;
            m.internal_medium = pm.internal_medium;
            m.external_medium = pm.external_medium;
            s->materials.push_back(m);
        }
        for (int i = 0; i < d->n_objects; i++) {
            const PtkObject& po = d->objects[i];
            Object o;
            o.type = po.type;
            o.material = po.material;
            switch (po.type) {
                case PTK_OBJ_SPHERE: // Sphere::Sphere src/objects.cpp:90-94
                    o.position = V3(po.position);
                    o.radius = po.radius;
                    o.area = 4 * M_PI * o.radius * o.radius;
                    break;
                case PTK_OBJ_PLANE:
                case PTK_OBJ_RECTANGLE: { // Plane::Plane :170-176, Rectangle::Rectangle :223-229
                    o.position = V3(po.position);
                    o.v1 = normalize(V3(po.v1));
                    o.v2 = normalize(V3(po.v2));
                    o.normal = normalize(cross(o.v1, o.v2));
                    if (po.type == PTK_OBJ_RECTANGLE) {
                        o.L1 = po.L1;
                        o.L2 = po.L2;
                        o.area = o.L1 * o.L2;
                    }
                    break;
                }
                case PTK_OBJ_TRIANGLE:
                    o.tri = make_triangle(V3(po.position), V3(po.v1), V3(po.v2));
                    o.area = o.tri.area;
                    break;
                case PTK_OBJ_MESH: { // load_object_model + ObjectUnion::ObjectUnion src/objectunion.cpp:3-44
                    Mesh mesh;
                    std::vector<Tri> file_order;
                    for (int64_t t = 0; t < po.n_triangles; t++) {
                        int64_t g = po.first_triangle + t;
                        const double* p = d->tri_vertices + 9 * g;
                        Tri tri = make_triangle(V3(p), V3(p + 3), V3(p + 6));
                        if (d->tri_flags[g] & PTK_TRI_HAS_UV) { // set_vertex_UV :321-325
                            const double* uv = d->tri_uv + 6 * g;
                            tri.uv1 = V3(uv[0], uv[1], 0);
                            tri.uv2 = V3(uv[2], uv[3], 0);
                            tri.uv3 = V3(uv[4], uv[5], 0);
                        }
                        if (d->tri_flags[g] & PTK_TRI_SMOOTH) { // set_vertex_normals :327-332
                            const double* n = d->tri_normals + 9 * g;
                            tri.n1 = V3(n);
                            tri.n2 = V3(n + 3);
                            tri.n3 = V3(n + 6);
                            tri.smooth = true;
                        }
                        tri.face_index = t;
                        file_order.push_back(tri);
                    }
                    const bool light = s->materials[static_cast<size_t>(po.material)].is_light;
                    mesh.area = 0;
                    for (const Tri& t : file_order) mesh.area += t.area;
                    if (light) {
                        mesh.n_lights = file_order.size();
                        for (size_t j = 0; j < file_order.size(); j++) {
                            double c = file_order[j].area;
                            if (j != 0) c += mesh.cumulative_area[j - 1];
                            mesh.cumulative_area.push_back(c);
                            mesh.light_index.push_back(static_cast<int>(j));
                        }
                    }
                    std::vector<int> ids(file_order.size());
                    for (size_t j = 0; j < ids.size(); j++) ids[j] = static_cast<int>(j);
                    // the root sorts the mesh's own array; primitive_ID is the index afterwards
                    if (!ids.empty()) {
                        // build with ids referring to file order, then renumber to sorted order
                        Mesh tmp;
                        build_node(tmp, file_order, ids.data(), static_cast<int>(ids.size()), d->settings.bvh_leaf_size,
                                   d->settings.bvh_n_axis_splits);
                        std::vector<int> new_id(ids.size());
                        for (size_t k = 0; k < ids.size(); k++) new_id[static_cast<size_t>(ids[k])] = static_cast<int>(k);
                        mesh.tris.resize(ids.size());
                        for (size_t k = 0; k < ids.size(); k++) mesh.tris[k] = file_order[static_cast<size_t>(ids[k])];
                        mesh.nodes = std::move(tmp.nodes);
                        for (BvhNode& node : mesh.nodes)
                            for (int& p : node.prims) p = new_id[static_cast<size_t>(p)];
                    }
                    else {
                        Mesh tmp;
                        build_node(tmp, file_order, ids.data(), 0, d->settings.bvh_leaf_size, d->settings.bvh_n_axis_splits);
                        mesh.nodes = std::move(tmp.nodes);
                    }
                    o.area = mesh.area;
                    o.mesh = static_cast<int>(s->meshes.size());
                    s->meshes.push_back(std::move(mesh));
                    break;
                }
                default: return set_error(PTK_ERR_INVALID, "oracle_scene_create: unknown object type");
            }
            s->objects.push_back(o);
        }
        s->background_medium = d->background_medium;
        s->background_color = V3(d->background_color);
        // Camera::Camera src/camera.cpp:4-22
        const PtkCamera& c = d->camera;
        s->cam_position = V3(c.position);
        s->cam_view = rotate(V3(0, -1, 0), c.orientation_y, c.orientation_z, c.orientation_x);
        s->cam_screen_y = rotate(V3(0, 0, -1), c.orientation_y, c.orientation_z, c.orientation_x);
        s->cam_width = c.camera_width;
        s->cam_focal = c.focal_length;
        s->cam_dof = c.dof_mode;
        s->cam_aperture = c.aperture_size;
        s->cam_focus = c.focus_distance;
        s->cam_screen_height = c.camera_width * static_cast<double>(d->settings.height) / static_cast<double>(d->settings.width);
        s->cam_screen_x = cross(s->cam_view, s->cam_screen_y);
        *out = s.release();
        return PTK_OK;
    }
    catch (const std::exception& e) {
        return set_error(PTK_ERR_INVALID, e.what());
    }
}

void oracle_scene_destroy(OracleScene* scene) {
    delete scene;
}

int oracle_mesh_face_index(const OracleScene* scene, int32_t object, int64_t primitive_id, int64_t* face_index) {
    if (!scene || object < 0 || object >= static_cast<int32_t>(scene->objects.size()) || !face_index)
        return set_error(PTK_ERR_INVALID, "oracle_mesh_face_index: bad argument");
    const Object& o = scene->objects[static_cast<size_t>(object)];
    if (o.type != PTK_OBJ_MESH) {
        *face_index = 0;
        return PTK_OK;
    }
    const Mesh& mesh = scene->meshes[static_cast<size_t>(o.mesh)];
    if (primitive_id < 0 || primitive_id >= static_cast<int64_t>(mesh.tris.size()))
        return set_error(PTK_ERR_INVALID, "oracle_mesh_face_index: primitive out of range");
    *face_index = mesh.tris[static_cast<size_t>(primitive_id)].face_index;
    return PTK_OK;
}

int oracle_trace_primary(OracleScene* scene, const PtkRenderJob* job, const OracleRng* rng_cfg, int32_t* object_index,
                         int32_t* primitive_index, double* distance, double* position, double* normal) {
    if (!check_job(scene, job)) return set_error(PTK_ERR_INVALID, "oracle_trace_primary: bad job");
    Ctx ctx;
    ctx.s = scene;
    init_rng(ctx.rng, rng_cfg);
    for (int64_t i = 0; i < job->number_of_pixels; i++) {
        int64_t idx = job->start_idx + i;
        size_t x, y;
        pixel_coordinates(*scene, idx, x, y);
        ctx.rng.start_path(job->seed, static_cast<uint32_t>(idx), static_cast<uint32_t>(job->sample_offset));
        Ray ray = camera_sample(*scene, x, y, ctx.rng);
        ray.t_max = INF;
        Hit hit;
        bool found = find_closest_hit(ctx, hit, ray);
        int face = 0;
        if (found && scene->objects[static_cast<size_t>(hit.object)].type == PTK_OBJ_MESH) {
            const Mesh& mesh = scene->meshes[static_cast<size_t>(scene->objects[static_cast<size_t>(hit.object)].mesh)];
            face = static_cast<int>(mesh.tris[static_cast<size_t>(hit.primitive)].face_index);
        }
        if (object_index) object_index[i] = found ? hit.object : -1;
        if (primitive_index) primitive_index[i] = found ? face : 0;
        if (distance) distance[i] = found ? hit.distance : INF;
        for (int j = 0; j < 3; j++) {
            if (position) position[3 * i + j] = found ? hit.point[j] : 0.0;
            if (normal) normal[3 * i + j] = found ? hit.normal[j] : 0.0;
        }
    }
    merge_stats(scene, ctx.stats);
    return PTK_OK;
}

int oracle_trace_samples(OracleScene* scene, const PtkRenderJob* job, const OracleRng* rng_cfg, double* color,
                         double* position, double* normal) {
    if (!check_job(scene, job)) return set_error(PTK_ERR_INVALID, "oracle_trace_samples: bad job");
    Ctx ctx;
    ctx.s = scene;
    init_rng(ctx.rng, rng_cfg);
    for (int64_t i = 0; i < job->number_of_pixels; i++) {
        int64_t idx = job->start_idx + i;
        size_t x, y;
        pixel_coordinates(*scene, idx, x, y);
        for (int32_t sidx = 0; sidx < job->number_of_samples; sidx++) {
            ctx.rng.start_path(job->seed, static_cast<uint32_t>(idx), static_cast<uint32_t>(job->sample_offset + sidx));
            Ray ray = camera_sample(*scene, x, y, ctx.rng);
            PixelData d = raytrace(ctx, ray);
            int64_t o = 3 * (i * job->number_of_samples + sidx);
            for (int j = 0; j < 3; j++) {
                if (color) color[o + j] = d.color[j];
                if (position) position[o + j] = d.position[j];
                if (normal) normal[o + j] = d.normal[j];
            }
        }
    }
    merge_stats(scene, ctx.stats);
    return PTK_OK;
}

int oracle_render(OracleScene* scene, const PtkRenderJob* job, const OracleRng* rng_cfg, int32_t threads, double* image,
                  double* position, double* normal, uint64_t* counters) {
    if (!check_job(scene, job) || !image || !position || !normal || !counters)
        return set_error(PTK_ERR_INVALID, "oracle_render: bad argument");
    const int mode = rng_cfg ? rng_cfg->mode : ORACLE_RNG_COUNTER;
    if (mode == ORACLE_RNG_REFERENCE_STREAMS || threads < 1) threads = 1;
    auto worker = [&](int64_t begin, int64_t end) {
        Ctx ctx;
        ctx.s = scene;
        init_rng(ctx.rng, rng_cfg);
        for (int64_t i = begin; i < end; i++) {
            int64_t idx = job->start_idx + i;
            size_t x, y;
            pixel_coordinates(*scene, idx, x, y);
            // compute_pixel_color app/pathtracer.cpp:178-201
            PixelData sum;
            for (int32_t sidx = 0; sidx < job->number_of_samples; sidx++) {
                ctx.rng.start_path(job->seed, static_cast<uint32_t>(idx), static_cast<uint32_t>(job->sample_offset + sidx));
                Ray ray = camera_sample(*scene, x, y, ctx.rng);
                PixelData d = raytrace(ctx, ray);
                sum.position += d.position;
                sum.normal += d.normal;
                sum.color += d.color;
            }
            double n = static_cast<double>(job->number_of_samples);
            V3 c = sum.color / n, p = sum.position / n, nn = sum.normal / n;
            double prev = static_cast<double>(counters[idx]);
            for (int j = 0; j < 3; j++) {
                image[3 * idx + j] = increment_average(image[3 * idx + j], c[j], prev, n);
                position[3 * idx + j] = increment_average(position[3 * idx + j], p[j], prev, n);
                normal[3 * idx + j] = increment_average(normal[3 * idx + j], nn[j], prev, n);
            }
            counters[idx] += static_cast<uint64_t>(job->number_of_samples);
        }
        merge_stats(scene, ctx.stats);
    };
    if (threads == 1) {
        worker(0, job->number_of_pixels);
    }
    else {
        std::vector<std::thread> pool;
        int64_t chunk = (job->number_of_pixels + threads - 1) / threads;
        for (int t = 0; t < threads; t++) {
            int64_t b = t * chunk, e = std::min<int64_t>(job->number_of_pixels, b + chunk);
            if (b < e) pool.emplace_back(worker, b, e);
        }
        for (auto& th : pool) th.join();
    }
    return PTK_OK;
}

int oracle_get_stats(OracleScene* scene, OracleStats* stats) {
    if (!scene || !stats) return set_error(PTK_ERR_INVALID, "oracle_get_stats: null argument");
    std::lock_guard<std::mutex> lock(scene->stats_mutex);
    *stats = scene->stats;
    return PTK_OK;
}

// ---------------------------------------------------------------------------------
// denoise (src/denoise.cpp)
// ---------------------------------------------------------------------------------
namespace {
int reflect_coordinate(int x, int size) { // clamp_x_coordinate / clamp_y_coordinate :20-38
    if (x < 0) return -x;
    else if (x > size - 1) return 2 * (size - 1) - x;
    return x;
}
double weight_component(const V3& p, const V3& q, double standard_deviation) { // :40-45
    if (standard_deviation == 0) return 0;
    return std::exp(-len(p - q) / (standard_deviation * standard_deviation));
}
int expand_kernel_idx(int idx, int hole_width) { // :68-82
    if (idx == 1) return idx + hole_width;
    else if (idx == -1) return idx - hole_width;
    else if (idx == 2) return idx + 2 * hole_width;
    else if (idx == -2) return idx - 2 * hole_width;
    return 0;
}
const double ATROUS_KERNEL[25] = {1.0 / 16.0, 1.0 / 16.0, 1.0 / 16.0, 1.0 / 16.0, 1.0 / 16.0, 1.0 / 16.0, 1.0 / 4.0,
                                  1.0 / 4.0,  1.0 / 4.0,  1.0 / 16.0, 1.0 / 16.0, 1.0 / 4.0,  3.0 / 8.0,  1.0 / 4.0,
                                  1.0 / 16.0, 1.0 / 16.0, 1.0 / 4.0,  1.0 / 4.0,  1.0 / 4.0,  1.0 / 16.0, 1.0 / 16.0,
                                  1.0 / 16.0, 1.0 / 16.0, 1.0 / 16.0, 1.0 / 16.0}; // include/denoise.h:11-14

void atrous_iteration(int W, int H, double* image, const double* position, const double* normal, double sigma_rt,
                      double sigma_x, double sigma_n, int hole_width) { // one_denoising_iteration :118-133
    std::vector<double> tmp(static_cast<size_t>(W) * H * 3);
    for (size_t p = 0; p < static_cast<size_t>(W) * H; p++) { // blur_pixel :84-116
        V3 new_value(0, 0, 0);
        double normalization = 0;
        int gx = static_cast<int>(p % static_cast<size_t>(W)), gy = static_cast<int>(p / static_cast<size_t>(W));
        V3 pixel_p(image + 3 * p), pos_p(position + 3 * p), normal_p(normal + 3 * p);
        for (int dx = -2; dx <= 2; dx++) {
            for (int dy = -2; dy <= 2; dy++) {
                int ex = expand_kernel_idx(dx, hole_width), ey = expand_kernel_idx(dy, hole_width);
                size_t kernel_idx = static_cast<size_t>(5 * (dy + 2) + (dx + 2));
                int x = reflect_coordinate(gx + ex, W), y = reflect_coordinate(gy + ey, H);
                size_t q = static_cast<size_t>(W) * static_cast<size_t>(y) + static_cast<size_t>(x);
                V3 pixel_q(image + 3 * q), pos_q(position + 3 * q), normal_q(normal + 3 * q);
                double w_rt = weight_component(pixel_p, pixel_q, sigma_rt); // compute_weight :47-66
                double w_x = weight_component(pos_p, pos_q, sigma_x);
                double w_n = weight_component(normal_p, normal_q, sigma_n);
                double weight = w_rt * w_x * w_n;
                double kernel_value = ATROUS_KERNEL[kernel_idx];
                V3 contribution = kernel_value * pixel_q * weight;
                if (std::isnan(len2(contribution))) {
                    contribution = V3(0.0);
                    weight = 0.0;
                }
                new_value += contribution;
                normalization += kernel_value * weight;
            }
        }
        V3 blurred = new_value / normalization;
        tmp[3 * p] = blurred.x;
        tmp[3 * p + 1] = blurred.y;
        tmp[3 * p + 2] = blurred.z;
    }
    std::memcpy(image, tmp.data(), tmp.size() * sizeof(double));
}

void median_filter(int W, int H, double* image, int kernel_size, double threshold) { // :149-201
    std::vector<double> tmp(static_cast<size_t>(W) * H * 3);
    int offset = (kernel_size - 1) / 2;
    size_t size = static_cast<size_t>(kernel_size) * static_cast<size_t>(kernel_size);
    std::vector<double> r(size), g(size), b(size);
    for (size_t idx = 0; idx < static_cast<size_t>(W) * H; idx++) {
        int x = static_cast<int>(idx % static_cast<size_t>(W)), y = static_cast<int>(idx / static_cast<size_t>(W));
        for (int dx = -offset; dx <= offset; dx++) {
            for (int dy = -offset; dy <= offset; dy++) {
                int lx = reflect_coordinate(x + dx, W), ly = reflect_coordinate(y + dy, H);
                size_t neighbor = static_cast<size_t>(W) * static_cast<size_t>(ly) + static_cast<size_t>(lx);
                size_t local = static_cast<size_t>(kernel_size) * static_cast<size_t>(dy + offset) + static_cast<size_t>(dx + offset);
                r[local] = image[3 * neighbor];
                g[local] = image[3 * neighbor + 1];
                b[local] = image[3 * neighbor + 2];
            }
        }
        std::nth_element(r.begin(), r.begin() + static_cast<long>(size / 2), r.end());
        std::nth_element(g.begin(), g.begin() + static_cast<long>(size / 2), g.end());
        std::nth_element(b.begin(), b.begin() + static_cast<long>(size / 2), b.end());
        V3 median(r[size / 2], g[size / 2], b[size / 2]);
        V3 colour(image + 3 * idx);
        if (len(colour - median) > threshold) colour = median;
        tmp[3 * idx] = colour.x;
        tmp[3 * idx + 1] = colour.y;
        tmp[3 * idx + 2] = colour.z;
    }
    std::memcpy(image, tmp.data(), tmp.size() * sizeof(double));
}
} // namespace

int oracle_denoise(const PtkDenoiseTask* tasks, int32_t n_tasks, int32_t W, int32_t H, double* image,
                   const double* position, const double* normal) { // denoise :203-212
    if ((!tasks && n_tasks > 0) || W < 1 || H < 1 || !image || !position || !normal)
        return set_error(PTK_ERR_INVALID, "oracle_denoise: bad argument");
    for (int32_t t = 0; t < n_tasks; t++) {
        const PtkDenoiseTask& task = tasks[t];
        if (task.mode == PTK_DENOISE_ATROUS) { // atrous_filter :135-147
            double sigma_rt = task.sigma_rt, sigma_x = task.sigma_x, sigma_n = task.sigma_n;
            int hole_width = 0;
            for (int it = 0; it < task.iterations; it++) {
                atrous_iteration(W, H, image, position, normal, sigma_rt, sigma_x, sigma_n, hole_width);
                sigma_rt /= 2.0;
                sigma_x /= 2.0;
                sigma_n /= 2.0;
                hole_width += static_cast<int>(std::pow(2, it));
            }
        }
        else if (task.mode == PTK_DENOISE_MEDIAN) {
            if (task.kernel_size < 1 || task.kernel_size % 2 == 0)
                return set_error(PTK_ERR_INVALID, "oracle_denoise: median kernel_size must be odd (the reference reads "
                                                  "uninitialised window entries for even sizes)");
            median_filter(W, H, image, task.kernel_size, task.threshold);
        }
    }
    return PTK_OK;
}

// ---------------------------------------------------------------------------------
// helper exports
// ---------------------------------------------------------------------------------
uint64_t oracle_rng_key(uint64_t seed, uint32_t pixel, uint32_t sample) { return rng_key(seed, pixel, sample); }
double oracle_rng_uniform(uint64_t key, uint32_t counter) { return rng_uniform(key, counter); }
uint64_t oracle_rng_fork(uint64_t key, uint32_t counter) { return rng_fork(key, counter); }

double oracle_fresnel_dielectric(double c, double n1, double n2) { return fresnel_dielectric(c, n1, n2); }
double oracle_fresnel_conductor(double c, double n1, double k1, double n2, double k2) {
    return fresnel_conductor(c, n1, k1, n2, k2);
}
double oracle_schlick_fresnel(double c, double n1, double n2) { return schlick_fresnel(c, n1, n2); }
double oracle_beckmann_D(const double h[3], const double n[3], double alpha) { return beckmann_D(V3(h), V3(n), alpha); }
double oracle_smith_G1(const double h[3], const double n[3], const double v[3], double alpha) {
    return smith_G1(V3(h), V3(n), V3(v), alpha);
}
void oracle_rotate(const double p[3], double a, double b, double g, double out[3]) {
    V3 r = rotate(V3(p), a, b, g);
    out[0] = r.x;
    out[1] = r.y;
    out[2] = r.z;
}
void oracle_refract(const double i[3], const double n[3], double eta, double out[3]) {
    V3 r = refract_vector(V3(i), V3(n), eta);
    out[0] = r.x;
    out[1] = r.y;
    out[2] = r.z;
}
int oracle_solve_quadratic(double b, double c, double* distance) {
    double d = 0;
    bool ok = solve_quadratic(b, c, d);
    if (ok && distance) *distance = d;
    return ok ? 1 : 0;
}

} // extern "C"
