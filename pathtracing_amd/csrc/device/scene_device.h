// Device-side scene tables and path-state pools (plain structs shared by host and kernels).
//
// Data layout in HBM
//   * scene tables (objects, materials, media, maps, lights): arrays of small structs, read with
//     wave-uniform indices in the object loop (scalar loads) or per-lane by material id;
//   * triangles: `tri_hot` holds the three vertices in fp64 (72 B, all the intersection test
//     reads), `tri_cold` everything shading needs; both in BVH leaf order;
//   * BVH: 4-wide nodes of 128 B with fp32 boxes rounded outward;
//   * path records: two pools of 16-byte planes used ping-pong (see PathPool);
//   * per-sample results indexed by the sample's batch index (see SampleResults).
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
    // first texel of each map (the whole map when it is 1x1): what shading reads when texture
    // coordinates are not in play, without walking maps -> map_data
    double c_albedo[3], c_emittance[3]; // emittance = emission colour * light intensity
    double c_roughness, c_transparency;
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
    int32_t index, _pad2;    // position in the scene's object table
    DevMaterial mat;         // copy of materials[material]: one load reaches everything shading needs
    double _pad3;            // 328 bytes: consecutive objects start 18 LDS banks apart, so lanes that read one
                             // field of different objects (8-byte reads) do not collide on a bank
};
static_assert(sizeof(DevObject) == 328, "the LDS copies of the object table rely on this stride");

// What the closest-hit loop (scan_objects, geometry.h) reads.  DevScene::scan points at a ScanHeader followed by the
// scene's objects in the order the loop takes them:
//   ShapeUnit[n_axis[0]]   rectangles whose normal is the x axis and whose sides run along y and z (up to sign):
//                          f[0..2] the rectangle's point, coordinates k, k + 1, k + 2 (cyclic; k the normal's axis),
//                          f[3], f[4] half sides + EPSILON along k + 1 and k + 2
//   ShapeUnit[n_axis[1]], ShapeUnit[n_axis[2]]   the same for y and z
//   ShapeUnit[n_spheres]   f[0..2] centre, f[3] radius
//   DevShape[n_others]     everything else in table order, meshes last (or every object in table order: ordered_scan)
// One type per loop: no dispatch on the type inside the loops, and 64 bytes (one scalar load) per object where that
// is enough.  The reference scans the objects in table order and the first one wins an exact tie; here the order is
// kept within each of the five runs, and between runs the earlier run wins.  Scenes in which that could matter
// (two flat objects in one plane that would fall into different runs) keep every object in the last run.
struct ShapeUnit { // 64 bytes
    double f[7];
    int32_t index; // the object's place in the scene's object table
    int32_t _pad;
};
struct ScanHeader { // 64 bytes
    int32_t n_axis[3];
    int32_t n_spheres;
    int32_t n_others;
    int32_t _pad[11];
};
static_assert(sizeof(ShapeUnit) == 64 && sizeof(ScanHeader) == 64, "the scan's records are read by 64-byte scalar loads");
// One object of the last run: 128 bytes fetched by two scalar loads.
//   sphere            f[0..2] centre, f[3] radius
//   plane, rectangle  f[0..2] point, f[3..5] normal, f[6..8] axis 1, f[9..11] axis 2, f[12] L1 / 2 + EPSILON, f[13] the
//                     same of L2
//   triangle          first_tri
//   mesh              first_tri, bvh_root, f[0..5] bounds of the whole mesh (lo xyz, hi xyz),
//                     f[6] != 0: no connection to a light passes through this mesh
struct DevShape {
    double f[14];
    int32_t type;      // PTK_OBJ_*
    int32_t index;     // the object's place in the scene's object table
    int32_t first_tri; // triangle: its index; mesh: first triangle of its range
    int32_t bvh_root;  // mesh: root node, -1 if empty
};
static_assert(sizeof(DevShape) == 128, "DevShape is read as two 64-byte scalar loads");

// The meshes of the scene that have triangles, in object order: what k_traverse needs to start a walk.
struct DevMeshRef { // 16 bytes
    int32_t object;    // index in the object table
    int32_t bvh_root;
    int32_t first_tri;
    int32_t opaque;    // no connection to a light passes through this mesh (DevShape::f[6])
};
constexpr int TRAVERSE_LDS_MESHES = 256; // entries of the list of meshes k_traverse keeps in LDS
constexpr int TRAVERSE_QUAD_LDS_MESHES = 128; // the same for k_traverse_quad

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

// Children per node: 4; 8 in measurement builds (-DPTK_BVH_WIDTH=8: host build and one-lane walk only).
#ifndef PTK_BVH_WIDTH
#define PTK_BVH_WIDTH 4
#endif
constexpr int BVH_WIDTH = PTK_BVH_WIDTH;
static_assert(BVH_WIDTH == 4 || BVH_WIDTH == 8, "nodes of four or eight children");
struct BvhNode4 { // 128 bytes (256 with eight children)
    float lo_x[BVH_WIDTH], lo_y[BVH_WIDTH], lo_z[BVH_WIDTH];
    float hi_x[BVH_WIDTH], hi_y[BVH_WIDTH], hi_z[BVH_WIDTH];
    int32_t child[BVH_WIDTH]; // >= 0 inner node; < 0 leaf: ~child = first_tri_local * 8 + (count - 1); EMPTY for none
    int32_t _pad[BVH_WIDTH];
};
constexpr int BVH_NODE_BYTES = 28 * BVH_WIDTH; // boxes and children, without the padding
constexpr int32_t BVH_EMPTY = INT32_MIN;
// The same node for k_traverse, which walks a hierarchy with four lanes per ray, one lane per child: child c's box
// and reference are 32 consecutive bytes (two 16-byte loads per lane; the four lanes of a ray read the four quarters
// of one 128-byte line with each of them).
struct BvhChildBox { // 32 bytes
    float lo[3], hi[3];
    int32_t child; // as BvhNode4::child
    int32_t _pad;
};
struct BvhNodeQuad { // 128 bytes
    BvhChildBox c[4];
};
// entries of the per-ray traversal stack of k_traverse (LDS); the builder is asked for trees that need fewer
#ifndef PTK_QUAD_STACK
#define PTK_QUAD_STACK 56
#endif
constexpr int QUAD_STACK_DEPTH = PTK_QUAD_STACK;
// traversal stack: entries kept in LDS per lane, and in a private spill array beyond that
#ifndef PTK_LDS_STACK
#define PTK_LDS_STACK 12
#endif
constexpr int LDS_STACK_DEPTH = PTK_LDS_STACK;
constexpr int SPILL_STACK_DEPTH = 64 - LDS_STACK_DEPTH;

struct DevCamera {
    double pos[3], view[3], screen_x[3], screen_y[3];
    double width, screen_height, focal, aperture, focus;
    int32_t dof_mode, _pad;
};

struct DevScene {
    int32_t n_objects, n_materials, n_media, n_maps;
    int32_t n_lights, has_triangles; // has_triangles: any triangle or mesh object
    int32_t has_meshes;
    int32_t needs_uv; // some material reads a map larger than 1x1
    int32_t has_pass_through;    // some material can let a connection to a light through (allow_direct_light)
    int32_t tables_fit_lds;      // objects + light_objects + media take at most LDS_SCENE_DOUBLES doubles
    int32_t has_axis_rectangles; // ScanHeader::n_axis is not all zero: rays carry the reciprocals of their direction
    int32_t has_transparency;    // some transparency map holds a value above zero (else no surface lets a ray straight through)
    int32_t width, height;
    int32_t max_depth, min_depth, nee, aa;
    int32_t background_medium, background_is_black;
    double background[3];
    DevCamera cam;
    const DevObject* objects;
    const ScanHeader* scan;      // the tables of the closest-hit loop (see ScanHeader)
    const DevMaterial* materials;
    const DevMedium* media;
    const DevMap* maps;
    const double* map_data;
    const int32_t* lights;       // object indices of light objects, in scene order
    const DevObject* light_objects; // copies of those objects, same order
    const TriHot* tri_hot;
    const TriCold* tri_cold;
    const BvhNode4* nodes;
    const BvhNodeQuad* nodes_quad;       // the same nodes, same indices, in the layout of k_traverse
    const double* light_cumulative_area; // per mesh light: cumulative areas in OBJ face order
    const int32_t* light_tri_index;      // per mesh light: root-sorted position -> triangle (leaf order)
    const DevMeshRef* meshes;            // [n_meshes]
    int32_t n_meshes, _pad2;
    uint32_t class_mask;                 // material classes present in the scene (+ scatter bit)
    float mesh_bound_abs;                // largest absolute coordinate of any mesh's bounds, rounded up
    uint32_t* stack_overflows;           // one counter: pushes a full traversal stack had to drop (TraversalStack::push)
};

// LDS every block of the shading kernels sets aside for copies of the object, light and medium tables
constexpr int LDS_SCENE_DOUBLES = 4096;

constexpr int N_STATISTICS = 8;         // values per statistics slot (see kernels.hip)
constexpr int STATISTICS_SLOTS = 8192; // per-block slots = the largest k_bounce grid

// Path records live in two pools used ping-pong.  A record is ten 16-byte elements ("planes", below).  A pool
// is laid out in tiles of RECORD_TILE records: within a tile, plane p of the 16 records is 256 consecutive
// bytes at p * 256, so a wave's load of one plane reads four runs of 256 bytes (whole cache lines), and all ten
// planes of a record are reached from one base pointer and one 32-bit offset, the plane being an immediate
// offset in the instruction.  Each bounce's kernels read the records of their class through the sorted index
// list and write the updated records to the other pool at the list position, so the records of the paths still
// in flight are always packed at the front of the current pool.
enum Plane : int {
    PL_RAY0 = 0, // origin.x, origin.y
    PL_RAY1,     // origin.z, direction.x
    PL_RAY2,     // direction.y, direction.z
    PL_THR0,     // throughput.x, throughput.y
    PL_THR1,     // throughput.z, scatter_pdf of the last non-virtual sampling event
    PL_SAV0,     // saved_point.x, saved_point.y (origin of that event)
    PL_SAV1,     // saved_point.z, radiance gathered so far: r
    PL_COL,      // radiance g, b
    PL_HIT,      // distance to the vertex to shade (double), flags (uint32), sample id (uint32)
    PL_IDS,      // uint32 x 4: next random-number counter, object index, triangle index (global leaf order), unused
    N_PLANES
};
constexpr uint32_t RECORD_TILE = 16;                                  // records per tile
constexpr uint32_t RECORD_PLANE_STRIDE = RECORD_TILE * 16;            // bytes between planes within a tile
constexpr uint32_t RECORD_TILE_BYTES = N_PLANES * RECORD_PLANE_STRIDE; // 2560
struct PathPool {
    char* base;   // capacity (a multiple of RECORD_TILE) * 160 bytes
    uint8_t* cls; // per record: class of the vertex to shade (0..7), or CLASS_DONE
};
// flags: depth (13 bits) | ray type << 13 | has_hit_surface << 15 | medium << 16

// Scenes with meshes trace in three passes per bounce.  k_bounce finishes every vertex whose shadow
// and extension rays miss the bounds of every mesh; the others it defers: it queues their rays
// (with t_max already cut to the closest hit among the other objects) and the vertex itself.
// k_traverse walks the hierarchies for the queued rays, one ray per lane, so that lanes and waves
// are filled with rays that really traverse; k_bounce_resume finishes the deferred vertices with the
// traversal results.  A deferred vertex is shaded once: k_bounce writes the record of the continuing path
// as far as it is known (everything but where the extension ray ends and the light the connection
// carries) and, beside it, what the rest needs (PENDING_PLANES planes indexed like the records).
struct RayRequest { // 64 bytes
    double ox, oy, oz, dx, dy, dz;
    double limit;    // t_max: the distance of the closest hit among the objects that are not meshes
    uint32_t slot;   // 2 * record position in the output pool + (0 shadow ray, 1 extension ray)
    uint32_t object; // that hit's object, 0xFFFFFFFF (-1) if none
};
struct RayResult { // 16 bytes: the closest hit of a deferred vertex's ray; obj < 0: none
    double t;
    int32_t obj, prim; // prim: the triangle when obj is a mesh
};
// What k_bounce_resume needs of a deferred vertex beyond its provisional record, 16-byte planes of [capacity]
// indexed by the record's position in the output pool:
//   0: sh_dir.x, sh_dir.y   1: sh_dir.z, sh_contrib.x   2: sh_contrib.y, sh_contrib.z   3: sh_dist, scatter_distance
//   4: sh_key, then two 32-bit words: light_index, and the vertex's medium (low 16 bits, MEDIUM_NONE for none) with
//      PENDING_CONNECT / PENDING_ALIVE
constexpr int PENDING_PLANES = 5;
constexpr uint32_t PENDING_CONNECT = 1u << 16, PENDING_ALIVE = 1u << 17;
struct DeferQueues {
    RayRequest* rays;
    RayResult* results;       // [2 * capacity], indexed by RayRequest::slot
    uint32_t* vertices;       // positions in the output pool of the deferred vertices
    uint32_t* counters;       // [4]: [0] rays queued, [1] vertices queued, [2] rays handed out by k_traverse
    double* pending;          // [PENDING_PLANES][capacity][2]
    uint64_t capacity;        // records per plane of `pending`
};

// Per-sample results (PixelData, app/pathtracer.cpp:20-24), indexed by the sample's batch index
// sid = sample * pixels_in_batch + pixel, so neighbouring lanes hold neighbouring pixels.
struct SampleResults {
    double* color;          // [n][4]: r, g, b, 1.0 if the path shaded a surface (else 0.0); written once, when the path ends
    double* surface;        // 3 planes of [n][2]: (px, py), (pz, nx), (ny, nz) of the first surface hit (valid if color[.][3] != 0)
    uint64_t plane_doubles; // doubles per plane of `surface`
};
constexpr uint32_t DEPTH_MASK = 0x1FFFu;
constexpr uint32_t TYPE_SHIFT = 13;
constexpr uint32_t FLAG_HAS_HIT = 1u << 15;
constexpr uint32_t MEDIUM_NONE = 0x7FFFu;
constexpr uint8_t CLASS_DONE = 0xFF; // the path ended and its radiance is in the results

struct BatchParams {
    int64_t start_idx;    // first pixel of the batch (banded jobs: of the job)
    uint32_t n_pixels;    // pixels in this batch
    uint32_t n_samples;   // samples per pixel in this batch
    uint32_t sample_offset;
    uint32_t n_paths;     // n_pixels * n_samples
    uint64_t seed;
    // Banded jobs (ptk_render_bands): the job's pixels are runs of band_pixels pixels that begin band_stride pixels
    // apart; pixel p of the batch is pixel band_first + p of the job.  band_pixels == 0: one contiguous run.
    uint32_t band_pixels, band_stride, band_first, _pad;
};

// A bounce's records are binned by class without atomics: k_bin_count leaves per-block class
// counts, k_bin_scan turns them into offsets (and this summary), k_bin_scatter writes the
// class-sorted index list.  BIN_CHUNK consecutive records belong to one block in all three.
constexpr int BIN_BLOCK = 256;
#ifndef PTK_BIN_ROUNDS
#define PTK_BIN_ROUNDS 4
#endif
constexpr int BIN_ROUNDS = PTK_BIN_ROUNDS;
constexpr int BIN_CHUNK = BIN_BLOCK * BIN_ROUNDS;
struct BinInfo {
    uint32_t start[N_CLASSES]; // first entry of each class segment in the sorted list
    uint32_t count[N_CLASSES];
    uint32_t total;            // length of the sorted list = records the bounce writes to the other pool
    uint32_t cursor;           // next list entry k_tail hands out (k_bin_scan leaves it at zero)
    uint32_t _pad[14];
};

} // namespace ptk
