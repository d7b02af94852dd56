// Device inlines: fp64 3-vectors, the counter-based random-number generator of include/ptk.h,
// Fresnel terms, direction sampling.  Each function names the reference code whose result it
// reproduces (paths relative to the reference repository).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptk {

#define PTK_DEV __device__ __forceinline__
#define PTK_CONSTANT __attribute__((address_space(4)))

constexpr double PI = 3.14159265358979323846;
constexpr double INF = __builtin_huge_val();
constexpr double INV_PI = 0x1.45f306dc9c883p-2; // 1 / PI rounded to nearest

// Square root and division as the compiler expands them (v_rsq_f64 / v_rcp_f64 plus Newton-Raphson steps and a
// final residual correction), without the range scaling it wraps around them for operands or results near the
// ends of the exponent range: same instructions on the data path, hence the same bits, for every value that occurs
// here; 13 and 8 instructions instead of about 20 and 12.
// fsqrt: exact zeros and +infinity pass through, negative operands give NaN; operands below 2^-1022 are not
// supported.  fdiv: for divisors that are never zero, infinite or denormal (x / 0 gives NaN, not infinity).
__device__ __forceinline__ double fsqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
    g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
    return (x == 0.0 || x == __builtin_huge_val()) ? x : g;
}
// fdiv in two halves: the refined reciprocal of the divisor, and the quotient from it.  fdiv(a, b) ==
// fdiv_by(a, b, frcp(b)) bit for bit, so quotients that share a divisor share its reciprocal (the v_rcp_f64 at a
// quarter of the rate of an fma, and four of the eight instructions).  frcp(-b) == -frcp(b).
__device__ __forceinline__ double frcp(double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = __builtin_fma(r, __builtin_fma(-b, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-b, r, 1.0), r);
    return r;
}
__device__ __forceinline__ double fdiv_by(double a, double b, double r) {
    const double q = a * r;
    return __builtin_fma(__builtin_fma(-b, q, a), r, q);
}
__device__ __forceinline__ double fdiv(double a, double b) { return fdiv_by(a, b, frcp(b)); }
// fsqrt for operands that are neither zero nor infinite (a negative operand still gives NaN)
__device__ __forceinline__ double fsqrt_finite(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
    g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
    return g;
}

// x / PI, correctly rounded like the division itself: the quotient from the rounded reciprocal and one residual
// correction (three instructions; the IEEE division expands to about fourteen).  Checked against x / PI on 4e8 values
// of x between 2^-60 and 1 without a mismatch; for x whose quotient is neither subnormal nor infinite.
__device__ __forceinline__ double div_pi(double x) {
    const double q = x * INV_PI;
    return __builtin_fma(__builtin_fma(-3.14159265358979323846, q, x), INV_PI, q);
}

struct V3 {
    double x, y, z;
};
PTK_DEV V3 v3(double x, double y, double z) { return V3{x, y, z}; }
PTK_DEV V3 v3(double s) { return V3{s, s, s}; }
PTK_DEV V3 v3(const double* p) { return V3{p[0], p[1], p[2]}; }
PTK_DEV V3 operator-(const V3& a) { return V3{-a.x, -a.y, -a.z}; }
PTK_DEV V3 operator+(const V3& a, const V3& b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
PTK_DEV V3 operator-(const V3& a, const V3& b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
PTK_DEV V3 operator*(const V3& a, const V3& b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
PTK_DEV V3 operator*(const V3& a, double s) { return V3{a.x * s, a.y * s, a.z * s}; }
PTK_DEV V3 operator*(double s, const V3& a) { return V3{a.x * s, a.y * s, a.z * s}; }
PTK_DEV V3 operator/(const V3& a, double s) { return V3{a.x / s, a.y / s, a.z / s}; }
// The library is built without automatic contraction (every kernel variant must round the same way), so
// fused multiply-adds are written out where they are wanted: sums of products cost 3 instructions instead
// of 5 and lose one rounding fewer than the reference's x86-64 build.
PTK_DEV double dot(const V3& a, const V3& b) { return fma(a.z, b.z, fma(a.y, b.y, a.x * b.x)); }
PTK_DEV V3 cross(const V3& a, const V3& b) {
    return V3{fma(a.y, b.z, -(a.z * b.y)), fma(a.z, b.x, -(a.x * b.z)), fma(a.x, b.y, -(a.y * b.x))};
}
PTK_DEV double len2(const V3& a) { return dot(a, a); }
// a + b * s
PTK_DEV V3 mad(const V3& b, double s, const V3& a) { return V3{fma(b.x, s, a.x), fma(b.y, s, a.y), fma(b.z, s, a.z)}; }
PTK_DEV double len(const V3& a) { return fsqrt(len2(a)); }
PTK_DEV V3 normalize(const V3& a) {
    const double l = len(a);
    return V3{fdiv(a.x, l), fdiv(a.y, l), fdiv(a.z, l)};
}
PTK_DEV double vmax(const V3& a) { return fmax(fmax(a.x, a.y), a.z); }
PTK_DEV V3 vexp(const V3& a) { return V3{exp(a.x), exp(a.y), exp(a.z)}; }
PTK_DEV double comp(const V3& a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
template<int K>
PTK_DEV double axis(const V3& a) { return K == 0 ? a.x : (K == 1 ? a.y : a.z); }

// ---- random numbers (contract in include/ptk.h) ------------------------------------------
constexpr uint64_t GOLDEN = 0x9E3779B97F4A7C15ull;
constexpr uint64_t FORK_SALT = 0xD1B54A32D192ED03ull;

PTK_DEV uint64_t mix64(uint64_t z) {
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}
PTK_DEV uint64_t rng_key(uint64_t seed, uint32_t pixel, uint32_t sample) {
    return mix64(mix64(seed + GOLDEN) ^ ((static_cast<uint64_t>(pixel) << 32) | sample));
}
PTK_DEV uint64_t rng_fork(uint64_t key, uint32_t n) {
    return mix64(key ^ mix64(static_cast<uint64_t>(n) + FORK_SALT));
}

struct Rng {
    uint64_t key;
    uint32_t ctr;
    PTK_DEV double u01() {
        uint64_t bits = mix64(key + (static_cast<uint64_t>(ctr) + 1) * GOLDEN);
        ctr++;
        return static_cast<double>(bits >> 11) * 0x1.0p-53;
    }
    PTK_DEV double uniform(double low, double high) { return (high - low) * u01() + low; }
    PTK_DEV int integer(int low, int high) { return static_cast<int>(uniform(low, high)); }
    PTK_DEV double normal() {
        double u1 = u01();
        double u2 = u01();
        return fsqrt(-2.0 * log(1.0 - u1)) * cos(2.0 * PI * u2);
    }
};

// ---- src/utils.cpp ------------------------------------------------------------------------
// fmod(x, 1.0) is x minus its integer part, exactly (the fractional part of a double is representable), so the two
// fmod calls of the reference cost a truncation and a subtraction each instead of the general remainder routine.
PTK_DEV double pos_fmod1(double a) { // pos_fmod(a, 1.0), :32-34
    const double shifted = (a - trunc(a)) + 1.0;
    return shifted - trunc(shifted);
}

PTK_DEV bool solve_quadratic(double b, double c, double& distance) { // :52-71
    double discriminant = b * b - 4.0 * c;
    if (discriminant <= 0) return false;
    double root = fsqrt(discriminant);
    double minimum_solution = -0.5 * (b + root);
    if (minimum_solution > EPSILON) {
        distance = minimum_solution;
        return true;
    }
    double maximum_solution = -0.5 * (b - root);
    if (maximum_solution > EPSILON) {
        distance = maximum_solution;
        return true;
    }
    return false;
}

// solve_quadratic without branches, so that the tests of two rays interleave: the square root of a
// negative discriminant is a NaN, which fails every comparison that follows.
PTK_DEV bool solve_quadratic_select(double b, double c, double& distance) {
    double discriminant = fma(b, b, -4.0 * c);
    double root = fsqrt_finite(discriminant); // (zero gives NaN here, and no root either way)
    double minimum_solution = -0.5 * (b + root), maximum_solution = -0.5 * (b - root);
    bool minimum_ok = minimum_solution > EPSILON, maximum_ok = maximum_solution > EPSILON;
    distance = minimum_ok ? minimum_solution : maximum_solution;
    return discriminant > 0 && (minimum_ok || maximum_ok);
}

PTK_DEV V3 sample_spherical(Rng& rng) { // :73-80
    double r1 = rng.normal();
    double r2 = rng.normal();
    double r3 = rng.normal();
    return normalize(v3(r1, r2, r3));
}

PTK_DEV void perpendicular_vectors(const V3& z_hat, V3& x_hat, V3& y_hat) { // :90-100
    V3 non_parallel = v3(1.0, 0.0, 0.0);
    if (fabs(z_hat.x) == 1.0) non_parallel = v3(0.0, 1.0, 0.0);
    x_hat = normalize(cross(z_hat, non_parallel));
    y_hat = normalize(cross(z_hat, x_hat));
}

PTK_DEV V3 sample_cosine_hemisphere(const V3& normal, Rng& rng) { // :115-126
    V3 x_hat, y_hat;
    perpendicular_vectors(normal, x_hat, y_hat);
    double theta = rng.u01() * 2 * PI;
    double radius = fsqrt(rng.u01());
    double s, c;
    sincos(theta, &s, &c);
    double x = c * radius;
    double y = s * radius;
    double z = fsqrt(1 - x * x - y * y);
    return mad(x_hat, x, mad(y_hat, y, normal * z));
}

PTK_DEV V3 reflect_vector(const V3& direction, const V3& normal) { // :128-130
    return mad(normal, -2.0 * dot(normal, direction), direction);
}

PTK_DEV V3 refract_vector(const V3& incident, const V3& normal, double eta) { // :132-141
    double cos_incident = dot(normal, incident);
    double k = 1 - eta * eta * (1 - cos_incident * cos_incident);
    if (k < 0) return v3(0.0);
    V3 perpendicular = incident - normal * cos_incident;
    return normal * fsqrt(k) + perpendicular * eta;
}

PTK_DEV double fresnel_dielectric(double cos_incident, double n1, double n2) { // :175-189
    double sin_incident = fsqrt(1 - cos_incident * cos_incident);
    double ratio = n1 / n2 * sin_incident;
    double cos_transmitted = fsqrt(1 - ratio * ratio);
    if (isnan(cos_transmitted)) return 0.0;
    double n1_ci = n1 * cos_incident, n2_ct = n2 * cos_transmitted;
    double n1_ct = n1 * cos_transmitted, n2_ci = n2 * cos_incident;
    double rs = (n1_ci - n2_ct) / (n1_ci + n2_ct);
    double rp = (n1_ct - n2_ci) / (n1_ct + n2_ci);
    return 0.5 * (rs * rs + rp * rp);
}

struct Cx {
    double re, im;
};
PTK_DEV Cx cx(double re, double im = 0.0) { return Cx{re, im}; }
PTK_DEV Cx operator+(Cx a, Cx b) { return Cx{a.re + b.re, a.im + b.im}; }
PTK_DEV Cx operator-(Cx a, Cx b) { return Cx{a.re - b.re, a.im - b.im}; }
PTK_DEV Cx operator*(Cx a, Cx b) { return Cx{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
PTK_DEV Cx operator*(double s, Cx a) { return Cx{s * a.re, s * a.im}; }
PTK_DEV Cx operator/(Cx a, double s) { return Cx{a.re / s, a.im / s}; }
PTK_DEV Cx operator/(Cx a, Cx b) { // Smith's algorithm
    if (fabs(b.re) < fabs(b.im)) {
        double ratio = b.re / b.im, denom = b.re * ratio + b.im;
        return Cx{(a.re * ratio + a.im) / denom, (a.im * ratio - a.re) / denom};
    }
    double ratio = b.im / b.re, denom = b.im * ratio + b.re;
    return Cx{(a.im * ratio + a.re) / denom, (a.im - a.re * ratio) / denom};
}
PTK_DEV Cx csqrt(Cx z) {
    if (z.re == 0.0 && z.im == 0.0) return Cx{0.0, z.im};
    double d = hypot(z.re, z.im);
    double r, s;
    if (z.re > 0) {
        r = fsqrt(0.5 * (d + z.re));
        s = 0.5 * (z.im / r);
    }
    else {
        s = fsqrt(0.5 * (d - z.re));
        r = fabs(0.5 * (z.im / s));
        s = copysign(s, z.im);
    }
    return Cx{r, s};
}

PTK_DEV double fresnel_conductor(double cos_theta_real, double n1, double k1, double n2, double k2) { // :191-223
    Cx cos_theta = cx(cos_theta_real);
    Cx one = cx(1.0);
    double eta, k;
    if (k1 == 0) {
        eta = n2 / n1;
        k = k2 / n1;
    }
    else {
        eta = n1 / n2;
        k = k1 / n2;
        Cx sin_theta = csqrt(one - cos_theta * cos_theta);
        Cx n_tilde = cx(n1, k1);
        Cx sin_theta_T = (n_tilde / n2) * sin_theta;
        Cx sin_theta_T2 = sin_theta_T * sin_theta_T;
        cos_theta = csqrt(one - sin_theta_T2);
    }
    Cx cos_theta2 = cos_theta * cos_theta;
    Cx sin_theta2 = one - cos_theta2;
    Cx base = cx(eta * eta - k * k) - sin_theta2;
    Cx f0 = csqrt(base * base + cx(4.0 * eta * eta * k * k));
    Cx a2b2 = f0;
    Cx a = csqrt(0.5 * f0 + cx(eta * eta - k * k) - sin_theta2);
    Cx f1 = a2b2 + cos_theta2;
    Cx f2 = 2.0 * (a * cos_theta);
    Cx f3 = cos_theta2 * a2b2 + sin_theta2 * sin_theta2;
    Cx f4 = f2 * sin_theta2;
    Cx R_s = (f1 - f2) / (f1 + f2);
    Cx R_p = R_s * (f3 - f4) / (f3 + f4);
    return 0.5 * (R_p.re + R_s.re);
}

PTK_DEV double schlick_fresnel(double cos_incident, double n1, double n2) { // :225-234
    double r0s = (n1 - n2) / (n1 + n2);
    double r0 = r0s * r0s;
    double x = 1 - cos_incident;
    double f = r0 + (1 - r0) * (x * x * x * x * x);
    if (isnan(f) || f < 0 || f > 1) return 0;
    return f;
}

PTK_DEV double fresnel_multiplier(double cos_incident, double n1, double k1, double n2, double k2, bool dielectric) { // :236-243
    if (dielectric || (k1 == 0 && k2 == 0)) return fresnel_dielectric(cos_incident, n1, n2);
    return fresnel_conductor(cos_incident, n1, k1, n2, k2);
}

// ---- microfacet helpers (src/materials.cpp:181-247) ---------------------------------------
PTK_DEV double chi(double x) { return x > 0 ? 1.0 : 0.0; }

PTK_DEV double beckmann_D(const V3& h, const V3& n, double alpha) { // :190-200
    double cos_theta = fmin(dot(h, n), 1.0);
    double c2 = cos_theta * cos_theta;
    double c4 = c2 * c2;
    double tan2 = (1.0 - c2) / c2; // the reference takes sqrt and squares again; same value up to rounding
    double a2 = alpha * alpha;
    double fraction = 1.0 / (PI * a2 * c4);
    return chi(cos_theta) * fraction * exp(-tan2 / a2);
}

PTK_DEV double smith_G1(const V3& h, const V3& n, const V3& v, double alpha) { // :202-212
    double cos_theta = dot(h, v);
    double tan_theta = fsqrt((1.0 - cos_theta * cos_theta) / (cos_theta * cos_theta));
    double a = 1.0 / (alpha * tan_theta);
    double approximation = a < 1.6 ? (3.535 * a + 2.181 * a * a) / (1.0 + 2.276 * a + 2.577 * a * a) : 1.0;
    return chi(cos_theta / dot(v, n)) * approximation;
}

PTK_DEV double smith_G(const V3& h, const V3& n, const V3& incident, const V3& outgoing, double alpha) { // :214-218
    return smith_G1(h, n, -incident, alpha) * smith_G1(h, n, outgoing, alpha);
}

PTK_DEV V3 sample_half_vector(const V3& normal, double alpha, Rng& rng) { // :220-235
    double r1 = rng.u01();
    double r2 = rng.u01();
    double phi = 2 * PI * r2;
    double tan_theta2 = -alpha * alpha * log(1 - r1);
    double cos_theta2 = 1.0 / (1.0 + tan_theta2);
    double cos_theta = fsqrt(cos_theta2);
    double sin_theta = fsqrt(1 - cos_theta2);
    V3 x_hat, y_hat;
    perpendicular_vectors(normal, x_hat, y_hat);
    double s, c;
    sincos(phi, &s, &c);
    return x_hat * sin_theta * c + y_hat * sin_theta * s + normal * cos_theta;
}

PTK_DEV double mis_weight(double pdf_a, double pdf_b) { // src/objects.cpp:464-468 with n_a = n_b = 1
    return pdf_a / (pdf_a + pdf_b); // (IEEE division: a pdf can be infinite, and x / inf is 0 where fdiv gives NaN)
}

} // namespace ptk
