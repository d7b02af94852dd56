// TEST INFRASTRUCTURE -- runs the reference's own asset loaders and hands back what they produced.
//
// Compiled with the reference's unmodified sources into oracle/_ref/libref.so (oracle/Makefile), in the build
// container only.  load_object_model (src/objectunion.cpp:425-443) and create_value_map_1D / _3D
// (src/valuemap.cpp:54-100) are called as they are; their results live in private members (Triangle's vertices,
// ObjectUnion's triangle array, ValueMap's texel array), which this one translation unit reads by compiling the
// reference's headers with the access specifiers neutralised.  That changes no layout and no code of the
// reference's own translation units; it only lets this file look.  The outputs become fixtures under tests/golden/
// (tests/golden/make_golden.py), against which the repository's own OBJ and .map readers are compared bit for bit.
// every standard header the reference's headers pull in comes first, with its access specifiers intact
#include <algorithm>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstring>
#include <fstream>
#include <iostream>
#include <mutex>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#define private public
#define protected public
#include "objects.h"
#include "objectunion.h"
#include "valuemap.h"
#undef private
#undef protected
#include "materials.h"

extern "C" {

// Runs load_object_model.  transform_flags: 1 center given (move), 2 orientation given (rotate), 4 size given (scale),
// as load_object (src/scene.cpp:383-412) would set them.  Returns the number of triangles, -1 when the reference threw
// (message in `error`), -2 when it produced a null triangle (it would crash on it later).  Arrays, when not null and
// large enough (capacity triangles): vertices [n][9], uv [n][6], normals [n][9], smooth [n].  The order is that of the
// ObjectUnion's array, i.e. after the root of its hierarchy has sorted it.
long ref_probe_load_object_model(const char* file, int smooth, int transform_flags, const double* center, const double* orientation,
                                 double size, long capacity, double* vertices, double* uv, double* normals, unsigned char* smooth_out,
                                 char* error, int error_size) {
    try {
        MaterialData data; // a default material: the loader only stores the pointer
        DiffuseMaterial* material = new DiffuseMaterial(data);
        ObjectTransform t;
        t.move_object = (transform_flags & 1) != 0;
        t.rotate_object = (transform_flags & 2) != 0;
        t.scale_object = (transform_flags & 4) != 0;
        t.center = center ? vec3(center[0], center[1], center[2]) : vec3(0, 0, 0);
        t.orientation = orientation ? vec3(orientation[0], orientation[1], orientation[2]) : vec3(0, 0, 0);
        t.size = size;
        // the reference's ObjectUnion constructor dereferences every entry: look at the triangle array first, through
        // the same functions load_object_model calls, when asked to (capacity < 0 means "only check for null entries")
        if (capacity < 0) {
            DataSizes nums = get_vertex_data_sizes(file);
            vec3* v = new vec3[nums.num_vertices];
            vec3* vt = new vec3[nums.num_vertex_UVs];
            vec3* vn = new vec3[nums.num_vertex_normals];
            populate_vertex_arrays(file, v, vt, vn);
            Object** triangles = new Object*[nums.num_triangles + 1]();
            int n = populate_triangle_array(file, v, vt, vn, triangles, material, smooth != 0);
            for (int i = 0; i < n; i++)
                if (!triangles[i]) return -2;
            return n;
        }
        ObjectUnion* u = load_object_model(file, material, smooth != 0, t);
        const long n = u->number_of_objects;
        if (n > capacity) return n;
        for (long i = 0; i < n; i++) {
            const Triangle* tri = static_cast<const Triangle*>(u->objects[i]);
            const vec3 p[3] = {tri->p1, tri->p2, tri->p3};
            const vec3 t3[3] = {tri->uv1, tri->uv2, tri->uv3};
            const vec3 n3[3] = {tri->n1, tri->n2, tri->n3};
            for (int k = 0; k < 3; k++)
                for (int a = 0; a < 3; a++) {
                    vertices[9 * i + 3 * k + a] = p[k][a];
                    normals[9 * i + 3 * k + a] = n3[k][a];
                    if (a < 2) uv[6 * i + 2 * k + a] = t3[k][a];
                }
            smooth_out[i] = tri->smooth_shaded ? 1 : 0;
        }
        return n;
    }
    catch (const std::exception& e) {
        if (error && error_size > 0) {
            std::strncpy(error, e.what(), static_cast<size_t>(error_size) - 1);
            error[error_size - 1] = 0;
        }
        return -1;
    }
}

// Runs create_value_map_1D (channels == 1) or create_value_map_3D (channels == 3, with the gamma flag).  Returns the
// number of values (width * height * channels), -1 when the reference threw.
long ref_probe_create_value_map(const char* file, int channels, int gamma_correct, long capacity, double* data, int* width, int* height,
                                char* error, int error_size) {
    try {
        const ValueMap* m = channels == 1 ? static_cast<const ValueMap*>(create_value_map_1D(file))
                                          : static_cast<const ValueMap*>(create_value_map_3D(file, gamma_correct != 0));
        *width = m->width;
        *height = m->height;
        const long n = static_cast<long>(m->width) * m->height * channels;
        if (n <= capacity) std::memcpy(data, m->data, static_cast<size_t>(n) * sizeof(double));
        return n;
    }
    catch (const std::exception& e) {
        if (error && error_size > 0) {
            std::strncpy(error, e.what(), static_cast<size_t>(error_size) - 1);
            error[error_size - 1] = 0;
        }
        return -1;
    }
}

} // extern "C"
