// Device-side scene tables and path-state pools (plain structs shared by host and kernels).
//
// Data layout in HBM
//   * scene tables (objects, materials, media, maps, lights): arrays of small structs, read with
//     wave-uniform indices in the object loop (scalar loads) or per-lane by material id;
//   * triangles: `tri_hot` holds the three vertices in fp64 (72 B, all the intersection test
//     reads), `tri_cold` everything shading needs; both in BVH leaf order;
//   * BVH: 4-wide nodes of 128 B with fp32 boxes rounded outward;
//   * path state: structure of arrays indexed by slot, two pools used ping-pong so that the
//     survivors of a bounce are written densely (compaction is fused into the shade kernels);
//   * per-sample results: structure of arrays indexed by the sample's batch index
//     sid = sample * pixels_in_batch + pixel, so neighbouring lanes hold neighbouring pixels.
#pragma once
#include <stdint.h>

#include "ptk.h"

namespace ptk {

constexpr double EPSILON = 0.000001; // the reference's constants::EPSILON (src/constants.cpp:22)

enum RayType : uint32_t { RAY_DIFFUSE = 0, RAY_REFLECTED = 1, RAY_TRANSMITTED = 2 }; // include/utils.h:14

// queue classes: one per material type, plus medium-scatter events
constexpr int CLASS_SCATTER = 7;
constexpr int N_CLASSES = 8;

struct DevMap {
    int32_t width, height, dim, _pad;
    int64_t offset; // into map_data
};

struct DevMedium {
    int32_t type;       // PTK_MEDIUM_*
    int32_t is_clear;   // extinction == 0 in every channel: transmittance is exactly 1
    double ior;
    double absorption[3], scattering[3], extinction[3], emission[3];
};

struct DevMaterial {
    int32_t type;
    int32_t is_light, is_dielectric;
    int32_t needs_uv; // some referenced map is larger than 1x1
    int32_t albedo, emission, intensity, roughness, transparency; // map ids, -1 = constant zero
    int32_t internal_medium, external_medium;                    // -1 = none
    int32_t _pad;
    double n, k;
};

struct DevObject {
    int32_t type, material;
    int32_t is_light, _pad;
    double px, py, pz;       // sphere centre / plane point
    double v1x, v1y, v1z;    // plane axes, normalised
    double v2x, v2y, v2z;
    double nx, ny, nz;       // plane normal
    double radius, L1, L2, area;
    int64_t first_tri;       // triangle (n_tri = 1) or mesh: range in the triangle arrays
    int64_t n_tri;
    int32_t bvh_root;        // mesh: index of its root node, -1 if the mesh is empty
    int32_t n_light_tris;    // mesh of a light material: number of triangles, else 0
    int64_t light_table;     // mesh light: offset of its cumulative-area / index tables
    double mesh_light_area;  // mesh light: last cumulative area
};

struct TriHot { // vertices, fp64
    double p1[3], p2[3], p3[3];
};

struct TriCold { // Triangle's derived members (include/objects.h:114-138)
    double normal[3], v1[3], v2[3];
    double x1, y1, x2, y2, x3, y3, det_T, area;
    double uv1[2], uv2[2], uv3[2];
    double n1[3], n2[3], n3[3];
    int32_t smooth;
    int32_t face_index; // index of the triangle in OBJ face order within its mesh
};

struct BvhNode4 { // 128 bytes
    float lo_x[4], lo_y[4], lo_z[4];
    float hi_x[4], hi_y[4], hi_z[4];
    int32_t child[4]; // >= 0 inner node; < 0 leaf: ~child = first_tri_local * 8 + (count - 1); EMPTY for none
    int32_t _pad[4];
};
constexpr int32_t BVH_EMPTY = INT32_MIN;
// traversal stack: entries kept in LDS per lane, and in a private spill array beyond that
constexpr int LDS_STACK_DEPTH = 24;
constexpr int SPILL_STACK_DEPTH = 40;

struct DevCamera {
    double pos[3], view[3], screen_x[3], screen_y[3];
    double width, screen_height, focal, aperture, focus;
    int32_t dof_mode, _pad;
};

struct DevScene {
    int32_t n_objects, n_materials, n_media, n_maps;
    int32_t n_lights, has_triangles;
    int32_t width, height;
    int32_t max_depth, min_depth, nee, aa;
    int32_t background_medium, background_is_black;
    double background[3];
    DevCamera cam;
    const DevObject* objects;
    const DevMaterial* materials;
    const DevMedium* media;
    const DevMap* maps;
    const double* map_data;
    const int32_t* lights;       // object indices of light objects, in scene order
    const TriHot* tri_hot;
    const TriCold* tri_cold;
    const BvhNode4* nodes;
    const double* light_cumulative_area; // per mesh light: cumulative areas in OBJ face order
    const int32_t* light_tri_index;      // per mesh light: root-sorted position -> triangle (leaf order)
    uint32_t class_mask;                 // material classes present in the scene (+ scatter bit)
};

// Path state: structure of arrays with one entry per sample of the batch.  The index of a sample,
// sid = sample * pixels_in_batch + pixel, is also its slot for the whole life of the path: nothing
// is ever moved.  Kernels reach the live paths through index lists (see kernels.hip).
struct PathState {
    double *ox, *oy, *oz;   // ray origin
    double *dx, *dy, *dz;   // ray direction
    double *tx, *ty, *tz;   // throughput
    double *sx, *sy, *sz;   // saved_point (origin of the last non-virtual sampling event)
    double *spdf;           // scatter_pdf of that event
    uint32_t* ctr;          // next random-number counter
    uint32_t* flags;        // depth (13 bits) | ray type << 13 | has_hit_surface << 15 | medium << 16
    // hit record: written by k_extend, read by k_shade
    double* hit_t;          // distance to the vertex (surface hit or scatter point)
    int32_t* hit_obj;       // object index
    int32_t* hit_prim;      // triangle index (global, leaf order) for triangles / meshes
    uint8_t* cls;           // queue class of the current vertex; CLASS_DEAD once the path has ended
    // pending next-event connection: written by k_shade, consumed by k_connect (origin = ox, oy, oz)
    double *sh_dx, *sh_dy, *sh_dz;
    double* sh_dist;           // distance to the sampled light point
    double *sh_cr, *sh_cg, *sh_cb; // contribution before emittance and transmittance
    uint64_t* sh_key;          // forked random-number key for the march
    uint32_t* sh_info;         // SHADOW_VALID | medium << 16 | light object index
    // results (PixelData, app/pathtracer.cpp:20-24)
    double *cr, *cg, *cb;      // radiance
    double *px, *py, *pz;      // first-surface position
    double *nx, *ny, *nz;      // first-surface normal
};
constexpr uint32_t DEPTH_MASK = 0x1FFFu;
constexpr uint32_t TYPE_SHIFT = 13;
constexpr uint32_t FLAG_HAS_HIT = 1u << 15;
constexpr uint32_t MEDIUM_NONE = 0x7FFFu;
constexpr uint32_t SHADOW_VALID = 1u << 31;
constexpr uint8_t CLASS_DEAD = 0xFF;

struct BatchParams {
    int64_t start_idx;    // first pixel
    uint32_t n_pixels;    // pixels in this batch
    uint32_t n_samples;   // samples per pixel in this batch
    uint32_t sample_offset;
    uint32_t n_paths;     // n_pixels * n_samples
    uint64_t seed;
};

// A bounce's list of live paths is binned by class without atomics: k_extend leaves per-block
// class counts, k_bin_scan turns them into offsets (and this summary), k_bin_scatter writes the
// class-sorted list.  BIN_CHUNK consecutive list entries belong to one block in all three.
constexpr int BIN_BLOCK = 256;
constexpr int BIN_ROUNDS = 8;
constexpr int BIN_CHUNK = BIN_BLOCK * BIN_ROUNDS;
struct BinInfo {
    uint32_t start[N_CLASSES]; // first entry of each class segment in the sorted list
    uint32_t count[N_CLASSES];
    uint32_t total;            // live paths = length of the sorted list
    uint32_t _pad[15];
};

} // namespace ptk
