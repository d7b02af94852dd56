// Device inlines: materials, media, light sampling, the visibility march.
// Reference code reproduced: src/materials.cpp (7 material types), src/medium.cpp,
// src/objects.cpp:68-166,257-265,408-412,444-580, src/objectunion.cpp:100-142.
#pragma once
#include "geometry.h"

namespace ptk {

struct Surface { // the reference's Hit (include/utils.h:22-30) plus texture coordinates
    V3 point, incident, normal;
    bool outside;
    int obj, prim;
    bool use_uv;
    double u, v;
};

struct BrdfSample { // BrdfData (include/materials.h:13-18)
    uint32_t type;
    double pdf;
    V3 outgoing, weight; // weight = brdf_over_pdf
};

PTK_DEV double medium_ior(const DevScene& sc, int medium) {
    return medium >= 0 ? sc.media[medium].ior : 1.0;
}

// Material::sample_transparency_map (src/materials.cpp:76-81): always consumes one number.
PTK_DEV bool sample_transparency(const DevScene& sc, const DevMaterial& m, const Surface& s, Rng& rng) {
    // No transparency map of the scene holds anything but zeros: the number is consumed but not computed.  (It would
    // let the ray through only if it were exactly zero, which one draw in 2^53 is.)
    if (!sc.has_transparency) {
        rng.ctr++;
        return false;
    }
    double transparency = s.use_uv ? tex1(sc, m.transparency, true, s.u, s.v) : m.c_transparency;
    return rng.u01() <= transparency;
}

PTK_DEV bool allow_direct_light(const DevScene& sc, const DevMaterial& m, const Surface& s, Rng& rng) {
    if (sample_transparency(sc, m, s, rng)) return true; // :55-60
    if (m.type == PTK_MAT_TRANSPARENT)                    // :129-136
        return medium_ior(sc, m.internal_medium) == medium_ior(sc, m.external_medium);
    return false;
}

PTK_DEV V3 material_emittance(const DevScene& sc, const DevMaterial& m, bool use_uv, double u, double v) { // :72-74
    if (!use_uv) return v3(m.c_emittance);
    return tex3(sc, m.emission, use_uv, u, v) * tex1(sc, m.intensity, use_uv, u, v);
}

PTK_DEV V3 material_albedo(const DevScene& sc, const DevMaterial& m, const Surface& s) {
    return s.use_uv ? tex3(sc, m.albedo, true, s.u, s.v) : v3(m.c_albedo);
}

PTK_DEV double get_alpha(const DevScene& sc, const DevMaterial& m, const Surface& s) { // :185-188
    return fmax(s.use_uv ? tex1(sc, m.roughness, true, s.u, s.v) : m.c_roughness, EPSILON);
}

PTK_DEV double specular_pdf(const DevScene& sc, const DevMaterial& m, const Surface& s, const V3& outgoing) { // :241-247
    V3 h = normalize(outgoing - s.incident);
    double half_pdf = beckmann_D(h, s.normal, get_alpha(sc, m, s)) * dot(h, s.normal);
    return fmax(half_pdf / (4.0 * dot(outgoing, h)), 0.0);
}

template<int T>
PTK_DEV double material_pdf(const DevScene& sc, const DevMaterial& m, const Surface& s, const V3& outgoing) { // brdf_pdf
    if (T == PTK_MAT_DIFFUSE) return div_pi(fmax(dot(outgoing, s.normal), 0.0)); // :103-106
    if (T == PTK_MAT_GLOSSY)                                                 // :316-320
        return 0.5 * (fmax(div_pi(dot(s.normal, outgoing)), 0.0) + specular_pdf(sc, m, s, outgoing));
    if (T == PTK_MAT_METALLIC_MICROFACET || T == PTK_MAT_REFLECTIVE_MICROFACET) return specular_pdf(sc, m, s, outgoing);
    return 0.0;
}

// The Fresnel-free part shared by the three reflective microfacet evals:
// D(h)(h.n) G / (-4 (i.n)(n.o)).
PTK_DEV double microfacet_core(const DevScene& sc, const DevMaterial& m, const Surface& s, const V3& h, const V3& outgoing) {
    double alpha = get_alpha(sc, m, s);
    double d_factor = beckmann_D(h, s.normal, alpha) * dot(h, s.normal);
    double g_factor = smith_G(h, s.normal, s.incident, outgoing, alpha);
    double denom_factor = -1.0 / (4.0 * dot(s.incident, s.normal) * dot(s.normal, outgoing));
    return d_factor * g_factor * denom_factor;
}

template<int T>
PTK_DEV V3 material_eval(const DevScene& sc, const DevMaterial& m, const Surface& s, const V3& outgoing, Rng& rng) {
    if (T == PTK_MAT_DIFFUSE) { // :83-88
        if (sample_transparency(sc, m, s, rng)) return v3(0.0);
        const V3 albedo = material_albedo(sc, m, s);
        return v3(div_pi(albedo.x), div_pi(albedo.y), div_pi(albedo.z));
    }
    if (T == PTK_MAT_GLOSSY) { // :249-286
        if (sample_transparency(sc, m, s, rng)) return v3(0.0);
        V3 h = normalize(outgoing - s.incident);
        double n1, n2;
        if (s.outside) {
            n1 = medium_ior(sc, m.external_medium);
            n2 = m.n;
        }
        else {
            n1 = m.n;
            n2 = medium_ior(sc, m.internal_medium);
        }
        double F_r = schlick_fresnel(-dot(s.incident, h), n1, n2);
        double r0s = (n1 - n2) / (n1 + n2);
        double R_0 = r0s * r0s;
        double fac1 = fmin(1.0 - dot(s.normal, -s.incident) / 2.0, 1.0);
        double fac2 = fmin(1.0 - dot(s.normal, outgoing) / 2.0, 1.0);
        V3 albedo = material_albedo(sc, m, s);
        V3 diffuse = albedo * 28.0 / (23.0 * PI) * (1 - R_0) * (1 - fac1 * fac1 * fac1 * fac1 * fac1) *
                     (1 - fac2 * fac2 * fac2 * fac2 * fac2);
        V3 reflection_color = m.is_dielectric ? v3(1.0) : albedo;
        return diffuse + reflection_color * (F_r * microfacet_core(sc, m, s, h, outgoing));
    }
    if (T == PTK_MAT_METALLIC_MICROFACET) { // :329-361 (no transparency test, no number consumed)
        V3 h = normalize(outgoing - s.incident);
        double n1, k1, n2, k2;
        if (s.outside) {
            n1 = medium_ior(sc, m.external_medium);
            k1 = 0;
            n2 = m.n;
            k2 = m.k;
        }
        else {
            n1 = m.n;
            k1 = m.k;
            n2 = medium_ior(sc, m.external_medium);
            k2 = 0;
        }
        double F_r = fresnel_multiplier(-dot(s.incident, h), n1, k1, n2, k2, false);
        return material_albedo(sc, m, s) * (F_r * microfacet_core(sc, m, s, h, outgoing));
    }
    if (T == PTK_MAT_REFLECTIVE_MICROFACET) { // :390-404
        if (sample_transparency(sc, m, s, rng)) return v3(0.0);
        V3 h = normalize(outgoing - s.incident);
        return material_albedo(sc, m, s) * microfacet_core(sc, m, s, h, outgoing);
    }
    return v3(0.0); // Reflective :108-110, Transparent :138-140, TransparentMicrofacet :406-408
}

template<int T>
PTK_DEV BrdfSample material_sample(const DevScene& sc, const DevMaterial& m, const Surface& s, Rng& rng) {
    BrdfSample r;
    if (sample_transparency(sc, m, s, rng)) { // sample_transparent_ray :45-52
        r.type = RAY_TRANSMITTED;
        r.outgoing = s.incident;
        r.weight = v3(1.0);
        r.pdf = 0;
        return r;
    }
    if (T == PTK_MAT_DIFFUSE) { // :90-101
        r.outgoing = sample_cosine_hemisphere(s.normal, rng);
        r.weight = material_albedo(sc, m, s);
        r.pdf = material_pdf<T>(sc, m, s, r.outgoing);
        r.type = RAY_DIFFUSE;
    }
    else if (T == PTK_MAT_REFLECTIVE) { // :112-123
        r.outgoing = reflect_vector(s.incident, s.normal);
        r.weight = m.is_dielectric ? v3(1.0) : material_albedo(sc, m, s);
        r.pdf = 0;
        r.type = RAY_REFLECTED;
    }
    else if (T == PTK_MAT_TRANSPARENT) { // :142-179
        double n1 = medium_ior(sc, s.outside ? m.external_medium : m.internal_medium);
        double n2 = medium_ior(sc, s.outside ? m.internal_medium : m.external_medium);
        V3 transmitted = refract_vector(s.incident, -s.normal, n1 / n2);
        double F_r = 1;
        if (len2(transmitted) != 0) F_r = fresnel_dielectric(-dot(s.incident, s.normal), n1, n2);
        bool reflected = rng.u01() <= F_r;
        r.type = reflected ? RAY_REFLECTED : RAY_TRANSMITTED;
        r.outgoing = reflected ? reflect_vector(s.incident, s.normal) : transmitted;
        r.weight = v3(1.0);
        r.pdf = 0;
    }
    else if (T == PTK_MAT_GLOSSY || T == PTK_MAT_METALLIC_MICROFACET || T == PTK_MAT_REFLECTIVE_MICROFACET) {
        // :288-314, :363-383 (ReflectiveMicrofacet inherits Metallic's sample)
        bool cosine_lobe = false;
        if (T == PTK_MAT_GLOSSY) cosine_lobe = rng.u01() <= 0.5;
        if (cosine_lobe) {
            r.outgoing = sample_cosine_hemisphere(s.normal, rng);
        }
        else {
            V3 h = sample_half_vector(s.normal, get_alpha(sc, m, s), rng);
            r.outgoing = reflect_vector(s.incident, h);
        }
        r.pdf = material_pdf<T>(sc, m, s, r.outgoing);
        if (r.pdf == 0) r.weight = v3(0.0);
        else r.weight = material_eval<T>(sc, m, s, r.outgoing, rng) * dot(r.outgoing, s.normal) / r.pdf;
        r.type = RAY_DIFFUSE;
    }
    else { // TransparentMicrofacet :410-455
        double n1 = medium_ior(sc, s.outside ? m.external_medium : m.internal_medium);
        double n2 = medium_ior(sc, s.outside ? m.internal_medium : m.external_medium);
        double alpha = get_alpha(sc, m, s);
        V3 h = sample_half_vector(s.normal, alpha, rng);
        double F_r = fresnel_dielectric(-dot(s.incident, h), n1, n2);
        V3 refracted = refract_vector(s.incident, -h, n1 / n2);
        double pick = rng.u01();
        if (pick <= F_r || len2(refracted) == 0) r.outgoing = reflect_vector(s.incident, h);
        else r.outgoing = refracted;
        r.type = RAY_TRANSMITTED;
        double cosine_factor = dot(s.incident, h) / (dot(s.incident, s.normal) * dot(h, s.normal));
        r.weight = v3(smith_G(h, s.normal, s.incident, r.outgoing, alpha) * cosine_factor);
        r.pdf = 0;
    }
    return r;
}

// ---- media (src/medium.cpp) -----------------------------------------------------------------
PTK_DEV V3 medium_transmittance(const DevMedium& m, double distance) { // :25-27
    if (m.is_clear) return v3(1.0);
    return vexp(v3(-m.extinction[0] * distance, -m.extinction[1] * distance, -m.extinction[2] * distance));
}

PTK_DEV V3 medium_sample_weight(const DevMedium& m, double distance, bool scatter) {
    V3 tr = medium_transmittance(m, distance);
    if (m.type == PTK_MEDIUM_BEERS_LAW) return tr; // :41-43
    V3 ext = v3(m.extinction);                      // :53-64
    V3 density = scatter ? ext * tr : tr;
    double pdf = (density.x + density.y + density.z) * (1.0 / 3.0);
    return scatter ? tr * v3(m.scattering) / pdf : tr / pdf;
}

PTK_DEV V3 medium_emission(const DevMedium& m) { // :66-73
    double pdf = (m.extinction[0] + m.extinction[1] + m.extinction[2]) * (1.0 / 3.0);
    return v3(m.emission) * v3(m.absorption) / pdf;
}

constexpr double PHASE_ISOTROPIC = 1.0 / (4 * PI); // :21-23

// ---- lights -----------------------------------------------------------------------------------
template<bool TRIS>
PTK_DEV double area_to_angle_factor(const DevScene& sc, const DevObject& o, const V3& surface_point, const V3& from,
                                    int prim) { // Object::area_to_angle_PDF_factor src/objects.cpp:68-76
    V3 normal = object_normal<TRIS>(sc, o, surface_point, prim);
    V3 difference = from - surface_point;
    return fabs(dot(normal, normalize(difference)) / len2(difference));
}

template<bool TRIS>
PTK_DEV double object_light_pdf(const DevScene& sc, const DevObject& o, const V3& surface_point, const V3& from, int prim) {
    if (o.type == PTK_OBJ_SPHERE) { // :130-138
        double distance = len(from - v3(o.px, o.py, o.pz));
        if (distance <= o.radius) return 1.0 / (o.area * area_to_angle_factor<TRIS>(sc, o, surface_point, from, 0));
        double q = o.radius / distance;
        double cos_theta_max = fsqrt(1 - q * q);
        return 1.0 / (2.0 * PI * (1 - cos_theta_max));
    }
    if (o.type == PTK_OBJ_PLANE) return 0.0; // :217-219
    if (TRIS && o.type == PTK_OBJ_MESH)      // src/objectunion.cpp:132-135
        return 1.0 / (o.mesh_light_area * area_to_angle_factor<TRIS>(sc, o, surface_point, from, prim));
    return fabs(1.0 / (o.area * area_to_angle_factor<TRIS>(sc, o, surface_point, from, prim))); // :78-80, :263-265
}

PTK_DEV V3 triangle_random_point(const TriHot& t, Rng& rng) { // :408-412
    double r1 = rng.u01();
    double r2 = rng.u01();
    double s = fsqrt(r1);
    return v3(t.p1) * (1.0 - s) + v3(t.p2) * (s * (1.0 - r2)) + v3(t.p3) * (s * r2);
}

// ObjectUnion::sample_random_primitive_index (src/objectunion.cpp:100-126).  The area table is in
// OBJ face order while the index it yields addresses the array the BVH root sorted; the second
// table maps that position to the triangle.
PTK_DEV int mesh_random_primitive(const DevScene& sc, const DevObject& o, Rng& rng) {
    const double* cumulative = sc.light_cumulative_area + o.light_table;
    double split = rng.uniform(0, o.area);
    int max = o.n_light_tris - 1, min = 0, index = 0;
    if (!(cumulative[0] >= split)) {
        while (min <= max) {
            index = (max + min) / 2;
            if (cumulative[index] < split) min = index + 1;
            else if (cumulative[index] == split || (cumulative[index] >= split && cumulative[index - 1] < split)) break;
            else max = index - 1;
        }
    }
    return sc.light_tri_index[o.light_table + index];
}

template<bool TRIS>
PTK_DEV V3 random_light_point(const DevScene& sc, const DevObject& o, const V3& from, double& pdf, Rng& rng) {
    if (o.type == PTK_OBJ_SPHERE) { // :140-166
        V3 centre = v3(o.px, o.py, o.pz);
        double distance = len(from - centre);
        if (distance <= o.radius) {
            V3 p = sample_spherical(rng) * o.radius + centre;
            pdf = object_light_pdf<TRIS>(sc, o, p, from, 0);
            return p;
        }
        double q = o.radius / distance;
        double cos_theta_max = fsqrt(1 - q * q);
        pdf = 1.0 / (2.0 * PI * (1 - cos_theta_max));
        double r = rng.u01();
        double cos_theta = 1 + r * (cos_theta_max - 1);
        double sin_theta = fsqrt(1 - cos_theta * cos_theta);
        double ds = distance * sin_theta;
        double inner = distance * cos_theta - fsqrt(o.radius * o.radius - ds * ds);
        double cos_alpha = (o.radius * o.radius + distance * distance - inner * inner) / (2.0 * distance * o.radius);
        double sin_alpha = fsqrt(1.0 - cos_alpha * cos_alpha);
        V3 x_hat, y_hat;
        V3 z_hat = normalize(from - centre);
        perpendicular_vectors(z_hat, x_hat, y_hat);
        double phi = rng.uniform(0, 2.0 * PI);
        double sp, cp;
        sincos(phi, &sp, &cp);
        V3 p = x_hat * sin_alpha * cp + y_hat * sin_alpha * sp + z_hat * cos_alpha;
        return p * o.radius + centre;
    }
    if (o.type == PTK_OBJ_RECTANGLE) { // :257-261
        double r1 = rng.uniform(-o.L1 / 2, o.L1 / 2);
        double r2 = rng.uniform(-o.L2 / 2, o.L2 / 2);
        V3 p = v3(o.v1x, o.v1y, o.v1z) * r1 + v3(o.v2x, o.v2y, o.v2z) * r2 + v3(o.px, o.py, o.pz);
        pdf = object_light_pdf<TRIS>(sc, o, p, from, 0);
        return p;
    }
    if (TRIS && o.type == PTK_OBJ_TRIANGLE) {
        V3 p = triangle_random_point(sc.tri_hot[o.first_tri], rng);
        pdf = object_light_pdf<TRIS>(sc, o, p, from, static_cast<int>(o.first_tri));
        return p;
    }
    if (TRIS && o.type == PTK_OBJ_MESH) { // src/objectunion.cpp:137-142
        int tri = mesh_random_primitive(sc, o, rng);
        V3 p = triangle_random_point(sc.tri_hot[tri], rng);
        pdf = object_light_pdf<TRIS>(sc, o, p, from, tri);
        return p;
    }
    pdf = 0; // plane: Object::random_light_point with the base surface point, pdf 0
    return v3(0.0);
}

// First half of sample_light (src/objects.cpp:515-551): everything before the visibility march.
// Returns false when the reference would return black without marching.
struct LightSample {
    int light_index;
    V3 direction;
    double distance, light_pdf;
};

// Rectangle, triangle and mesh lights: a uniform point on the surface (Rectangle::generate_random_surface_point
// src/objects.cpp:257-261, Triangle::generate_random_surface_point :408-412, ObjectUnion::random_light_point
// src/objectunion.cpp:137-142); prim: the triangle the point lies on.
template<bool TRIS>
PTK_DEV V3 random_area_point(const DevScene& sc, const DevObject& o, int& prim, Rng& rng) {
    prim = 0;
    if (TRIS && o.type == PTK_OBJ_TRIANGLE) {
        prim = static_cast<int>(o.first_tri);
        return triangle_random_point(sc.tri_hot[o.first_tri], rng);
    }
    if (TRIS && o.type == PTK_OBJ_MESH) {
        prim = mesh_random_primitive(sc, o, rng);
        return triangle_random_point(sc.tri_hot[prim], rng);
    }
    double r1 = rng.uniform(-o.L1 / 2, o.L1 / 2);
    double r2 = rng.uniform(-o.L2 / 2, o.L2 / 2);
    return v3(o.v1x, o.v1y, o.v1z) * r1 + v3(o.v2x, o.v2y, o.v2z) * r2 + v3(o.px, o.py, o.pz);
}

template<bool TRIS>
PTK_DEV bool pick_light_point(const DevScene& sc, int shaded_object, const V3& from, LightSample& ls, Rng& rng) {
    if (sc.n_lights == 0) return false;                                   // sample_random_light :444-462
    const DevObject& light = sc.light_objects[rng.integer(0, sc.n_lights)];
    ls.light_index = light.index;
    if (ls.light_index == shaded_object) return false;
    const bool by_area = light.type == PTK_OBJ_RECTANGLE || (TRIS && (light.type == PTK_OBJ_TRIANGLE || light.type == PTK_OBJ_MESH));
    V3 p;
    int prim = 0;
    if (by_area) p = random_area_point<TRIS>(sc, light, prim, rng);
    else {
        p = random_light_point<TRIS>(sc, light, from, ls.light_pdf, rng);
        if (ls.light_pdf == 0) return false;
    }
    V3 d = p - from;
    const double distance2 = len2(d);
    ls.distance = fsqrt(distance2);
    ls.direction = v3(fdiv(d.x, ls.distance), fdiv(d.y, ls.distance), fdiv(d.z, ls.distance));
    if (by_area) {
        // object_light_pdf for these shapes, sharing the unit vector and the squared distance with the connection:
        // area_to_angle_factor normalises from - p, whose components, length and quotients are those of p - from
        // with the sign changed, bit for bit, and the sign drops out under fabs.
        const V3 normal = object_normal<TRIS>(sc, light, p, prim);
        const double factor = fabs(fdiv(dot(normal, ls.direction), distance2));
        const double area = TRIS && light.type == PTK_OBJ_MESH ? light.mesh_light_area : light.area;
        ls.light_pdf = fabs(fdiv(1.0, area * factor));
        if (ls.light_pdf == 0) return false;
    }
    return true;
}

// compute_visibility (src/objects.cpp:470-513) plus sample_light's acceptance test (:558):
// marches towards the light through surfaces that let direct light pass, multiplying the
// transmittance of each segment's medium.  Returns emittance * transmittance, or black.
// `h` holds the closest hit of the first segment (found by the caller, together with the
// extension ray); later segments are traced here.
// PASS = false: no material of the scene lets a connection through, so the march is its first segment.
template<bool TRIS, bool UV, bool PASS>
PTK_DEV V3 march_to_light(const DevScene& sc, V3 org, const V3& dir, int medium, int light_index, double distance_to_light,
                          Rng& rng, TraversalStack& st, uint32_t& rays, HitInfo h) {
    V3 transmittance = v3(1.0);
    double distance = 0;
    for (;;) {
        if (h.obj < 0) return v3(0.0);
        distance += h.t;
        if (medium < 0) medium = sc.background_medium;
        transmittance = transmittance * medium_transmittance(sc.media[medium], h.t);
        const DevObject& o = sc.objects[h.obj];
        const DevMaterial& m = o.mat;
        V3 point = org + dir * h.t;
        Surface s;
        s.use_uv = UV && m.needs_uv != 0;
        s.u = s.v = 0;
        if (s.use_uv) object_uv<TRIS>(sc, o, point, h.prim, s.u, s.v);
        if (h.obj == light_index) {
            V3 emittance = material_emittance(sc, m, s.use_uv, s.u, s.v);
            if (fabs(distance_to_light - distance) > EPSILON) return v3(0.0);
            return emittance * transmittance;
        }
        if (!PASS || !allow_direct_light(sc, m, s, rng)) return v3(0.0);
        V3 normal = object_normal<TRIS>(sc, o, point, h.prim);
        bool outside = dot(dir, normal) < 0;
        org = point;
        medium = outside ? m.internal_medium : m.external_medium;
        rays++;
        closest_hit<TRIS>(sc, org, dir, INF, h, st);
    }
}

} // namespace ptk
