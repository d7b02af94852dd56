// Wavefront path-tracing kernels for gfx950.
//
//   k_generate     camera samples, their first closest hit -> path records
//                  (compute_pixel_color + Camera::make_ray + the first loop head of raytrace)
//   k_bin_scan     exclusive scan of the per-chunk class counts the last kernel left
//   k_bin_scatter  writes the class-sorted index list           (stable: record order is preserved)
//   k_bounce       one path vertex per lane, a block works through chunks of one material class:
//                  emission with MIS, next-event sampling and the visibility march of that
//                  connection (sample_light + compute_visibility), BRDF sampling, medium switch,
//                  Russian roulette (raytrace's loop body), then the free-flight distance and the
//                  closest hit of the new ray (the next loop head + find_closest_hit); reads the
//                  record through the list and writes the updated record, with the class of its
//                  next vertex, to the other pool at the list position; counts the classes it wrote
//   k_tail         the last bounces of a batch: one lane follows one path to its end, no sorting
//   k_accumulate   per-pixel sample means folded into the film  (compute_pixel_color's division and
//                  increment_average)
//
// One lane owns one path.  A bounce is one pass over the records: shading, the shadow ray and the
// extension ray happen while the record is in registers, so a vertex costs one read and one write
// of a 160-byte record.  Records move once per bounce, from one pool to the other, which keeps the
// records in flight packed (coalesced 16-byte accesses whatever fraction of the batch is still
// alive).  Sorting by class uses block-local ballots and a scan instead of global atomics: with one
// atomic per wave on a shared counter the kernels ran at the rate of that one cache line (about
// 100 atomics per microsecond) instead of at the rate of the arithmetic; the statistics counters
// are per-block slots for the same reason.  All arithmetic is fp64, as in the reference; MFMA has
// no role here (divergent scalar-per-lane work, no dense contraction).
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "shading.h"

namespace ptk {

constexpr int BLOCK = 256;
// Register budgets: the kernels are compiled for a fixed number of waves per SIMD (512 registers
// per lane and SIMD are shared by the resident waves).
#ifndef PTK_BOUNCE_WAVES
#define PTK_BOUNCE_WAVES 2
#endif
#ifndef PTK_BOUNCE_UV_WAVES
#define PTK_BOUNCE_UV_WAVES 2
#endif
#ifndef PTK_TRAVERSE_WAVES
#define PTK_TRAVERSE_WAVES 6
#endif
#define PTK_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n, n)))
static_assert(BLOCK == BIN_BLOCK, "k_extend and the bin kernels share their block geometry");

PTK_DEV TraversalStack make_stack(StackEntry* lds) {
    TraversalStack st;
    st.lds = (LdsStackEntry*) lds + threadIdx.x;
    st.stride = blockDim.x;
    st.sp = 0;
    return st;
}

PTK_DEV void sample_coordinates(const BatchParams& b, uint32_t sid, uint32_t& pixel, uint32_t& sample) {
    uint32_t s = sid / b.n_pixels;
    pixel = static_cast<uint32_t>(b.start_idx) + (sid - s * b.n_pixels);
    sample = b.sample_offset + s;
}

PTK_DEV uint32_t lanes_below(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0));
}

// Camera sample of one path: AA jitter (app/pathtracer.cpp:185-188), Camera::make_ray and
// adjust_depth_of_field (src/camera.cpp:24-70).
PTK_DEV void camera_ray(const DevScene& sc, uint32_t pixel, Rng& rng, V3& org, V3& dir) {
    const uint32_t W = static_cast<uint32_t>(sc.width), H = static_cast<uint32_t>(sc.height);
    double x = static_cast<double>(pixel % W);
    double y = static_cast<double>(H - pixel / W);
    if (sc.aa) {
        x += rng.normal() / 3.0;
        y += rng.normal() / 3.0;
    }
    const DevCamera& c = sc.cam;
    double local_x = x * c.width / static_cast<double>(W) - c.width / 2.0;
    double local_y = y * c.screen_height / static_cast<double>(H) - c.screen_height / 2.0;
    V3 position = v3(c.pos);
    V3 pixel_vector = position + v3(c.screen_x) * local_x + v3(c.screen_y) * local_y + v3(c.view) * c.focal;
    org = position;
    dir = normalize(pixel_vector - position);
    if (c.dof_mode != PTK_DOF_NONE) {
        double t = c.focus / dot(dir, v3(c.view));
        V3 focal_point = org + dir * t;
        V3 adjust;
        if (c.dof_mode == PTK_DOF_CIRCLE) {
            double r = c.aperture / 2.0 * sqrt(rng.u01());
            double phi = rng.uniform(0, 2 * PI);
            double s, co;
            sincos(phi, &s, &co);
            adjust = (v3(c.screen_x) * co + v3(c.screen_y) * s) * r;
        }
        else {
            double r1 = rng.uniform(0, c.aperture) - c.aperture / 2.0;
            double r2 = rng.uniform(0, c.aperture) - c.aperture / 2.0;
            adjust = v3(c.screen_y) * r1 + v3(c.screen_x) * r2;
        }
        org = org + adjust;
        dir = normalize(focal_point - org);
    }
}

// ---- record access ------------------------------------------------------------------------------
// A plane's address is uniform (scalar arithmetic); the record index enters as a 32-bit byte offset.
PTK_DEV char* plane_ptr(const PathPool& p, int plane) {
    return p.base + static_cast<uint64_t>(plane) * p.plane_bytes;
}
PTK_DEV double2 ld2(const PathPool& p, int plane, uint32_t i) {
    return *reinterpret_cast<const double2*>(plane_ptr(p, plane) + i * 16u);
}
PTK_DEV void st2(const PathPool& p, int plane, uint32_t i, double a, double b) {
    *reinterpret_cast<double2*>(plane_ptr(p, plane) + i * 16u) = make_double2(a, b);
}
PTK_DEV uint4 ldu4(const PathPool& p, int plane, uint32_t i) {
    return *reinterpret_cast<const uint4*>(plane_ptr(p, plane) + i * 16u);
}
PTK_DEV void stu4(const PathPool& p, int plane, uint32_t i, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    *reinterpret_cast<uint4*>(plane_ptr(p, plane) + i * 16u) = make_uint4(a, b, c, d);
}

// The path ended: its radiance goes to the per-sample results.
PTK_DEV void flush_color(const SampleResults& res, uint32_t sid, double r, double g, double bl) {
    double2* c = reinterpret_cast<double2*>(res.color) + 2 * static_cast<size_t>(sid);
    c[0] = make_double2(r, g);
    c[1] = make_double2(bl, 0.0);
}
PTK_DEV double2* surface_plane(const SampleResults& res, int plane) {
    return reinterpret_cast<double2*>(res.surface + static_cast<uint64_t>(plane) * res.plane_doubles);
}

// Statistics without atomics on shared lines: every block adds its totals to its own slot.
// values: [0] visibility-march segments, [1] extension rays, [2] connections marched,
// [3] records written for the next bounce, [4] vertices shaded by k_tail after its first round,
// [5] unused, [6] and [7] cycle stamps of the sections of k_bounce in the -DPTK_STAMPS build.
PTK_DEV void add_block_statistics(unsigned long long* slots, uint32_t (*wave_sums)[N_STATISTICS], uint32_t* mine) {
#pragma unroll
    for (int k = 0; k < N_STATISTICS; k++)
        for (int off = 32; off > 0; off >>= 1) mine[k] += __shfl_down(mine[k], off);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < N_STATISTICS; k++) wave_sums[threadIdx.x >> 6][k] = mine[k];
    }
    __syncthreads();
    if (threadIdx.x < N_STATISTICS) {
        uint32_t total = 0;
        for (int w = 0; w < BLOCK / 64; w++) total += wave_sums[w][threadIdx.x];
        if (total) slots[static_cast<size_t>(blockIdx.x % STATISTICS_SLOTS) * N_STATISTICS + threadIdx.x] += total;
    }
}

// -------------------------------------------------------------------------------------------------
// Loop head of raytrace (app/pathtracer.cpp:53-67) for a new ray: free-flight distance in a
// scattering medium, closest hit.  cls < 0: the ray left the scene.
struct Vertex {
    double t;
    int obj, prim, cls;
};

// HomogenousScatteringMedium::sample_distance (src/medium.cpp:45-51) for the medium the new ray
// starts in; infinity elsewhere.
PTK_DEV double free_flight_distance(const DevScene& sc, const BatchParams& b, uint32_t medium, uint32_t sid, uint32_t ctr) {
    if (medium == MEDIUM_NONE) medium = static_cast<uint32_t>(sc.background_medium);
    const DevMedium& med = sc.media[medium];
    double scatter_distance = INF;
    if (med.type == PTK_MEDIUM_HOMOGENEOUS_SCATTERING) {
        uint32_t pixel, sample;
        sample_coordinates(b, sid, pixel, sample);
        Rng rng{rng_key(b.seed, pixel, sample), ctr};
        int channel = rng.integer(0, 3);
        double ext = med.extinction[channel];
        if (ext != 0) scatter_distance = -log(rng.u01()) / ext;
    }
    return scatter_distance;
}

PTK_DEV Vertex vertex_of(const HitInfo& h, double scatter_distance) {
    Vertex v;
    if (h.obj >= 0) {
        v.t = h.t;
        v.obj = h.obj;
        v.prim = h.prim;
        v.cls = h.cls;
    }
    else { // a scatter event before any surface, or nothing at all
        v.t = scatter_distance;
        v.obj = -1;
        v.prim = 0;
        v.cls = scatter_distance == INF ? -1 : CLASS_SCATTER;
    }
    return v;
}

template<bool TRIS>
PTK_DEV Vertex next_vertex(const DevScene& sc, const BatchParams& b, const V3& org, const V3& dir, uint32_t medium, uint32_t sid,
                           uint32_t ctr, TraversalStack& st) {
    const double scatter_distance = free_flight_distance(sc, b, medium, sid, ctr);
    HitInfo h;
    closest_hit<TRIS>(sc, org, dir, scatter_distance, h, st);
    return vertex_of(h, scatter_distance);
}

// -------------------------------------------------------------------------------------------------
// Chunks.  The records of a pool lie class by class (the order the last bounce wrote them in); a
// chunk is up to rounds * BLOCK consecutive records of one class, where `rounds` (1..BIN_ROUNDS) is
// chosen by the host for every pool: large chunks amortise the per-chunk work, small ones keep
// every CU busy when few paths are left.  Chunks are numbered class by class,
// which is also the order of the records, and are the unit of work of k_bounce (one block works
// through one chunk at a time) and of the class sort (block_counts has one entry per class and
// chunk).
struct ChunkTable {
    uint32_t first_chunk[N_CLASSES + 1]; // chunks of class c: [first_chunk[c], first_chunk[c + 1])
};
PTK_DEV ChunkTable chunk_table(const BinInfo* info, uint32_t chunk_size) {
    ChunkTable t;
    uint32_t running = 0;
#pragma unroll
    for (int c = 0; c < N_CLASSES; c++) {
        t.first_chunk[c] = running;
        running += (info->count[c] + chunk_size - 1) / chunk_size;
    }
    t.first_chunk[N_CLASSES] = running;
    return t;
}
// class, first record and number of records of a chunk
PTK_DEV void locate_chunk(const BinInfo* info, const ChunkTable& t, uint32_t chunk_size, uint32_t chunk, int& cls, uint32_t& first,
                          uint32_t& count) {
    cls = 0;
#pragma unroll
    for (int c = 1; c < N_CLASSES; c++)
        if (chunk >= t.first_chunk[c]) cls = c;
    const uint32_t offset = (chunk - t.first_chunk[cls]) * chunk_size;
    first = info->start[cls] + offset;
    count = min(chunk_size, info->count[cls] - offset);
}

// Per-class counts of the records a block wrote for one chunk -> block_counts[class][chunk].
PTK_DEV void store_chunk_counts(uint32_t (*wave_counts)[N_CLASSES], const uint32_t* my_counts, uint32_t* block_counts,
                                uint32_t count_stride, uint32_t chunk) {
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int c = 0; c < N_CLASSES; c++) wave_counts[threadIdx.x >> 6][c] = my_counts[c];
    }
    __syncthreads();
    if (threadIdx.x < N_CLASSES) {
        uint32_t total = 0;
        for (int w = 0; w < BLOCK / 64; w++) total += wave_counts[w][threadIdx.x];
        block_counts[threadIdx.x * count_stride + chunk] = total;
    }
    __syncthreads(); // wave_counts is reused by the block's next chunk
}

// -------------------------------------------------------------------------------------------------
// Camera samples and their first vertex.  The first pool holds the records in sample order as one
// run (described to the class sort as class 0); chunk j = samples [j * BIN_CHUNK, (j + 1) * BIN_CHUNK).
template<bool TRIS>
__global__ __launch_bounds__(BLOCK) void k_generate(DevScene sc, BatchParams b, PathPool pool, SampleResults res, BinInfo* first_info,
                                                    uint32_t* block_counts, uint32_t count_stride) {
    __shared__ StackEntry lds_stack[TRIS ? LDS_STACK_DEPTH * BLOCK : 1];
    __shared__ uint32_t wave_counts[BLOCK / 64][N_CLASSES];
    TraversalStack st = make_stack(lds_stack);
    if (blockIdx.x == 0 && threadIdx.x < N_CLASSES) {
        first_info->start[threadIdx.x] = threadIdx.x == 0 ? 0u : b.n_paths;
        first_info->count[threadIdx.x] = threadIdx.x == 0 ? b.n_paths : 0u;
        if (threadIdx.x == 0) first_info->total = b.n_paths;
    }
    const uint32_t n_chunks = (b.n_paths + BIN_CHUNK - 1) / BIN_CHUNK;
    for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        uint32_t my_counts[N_CLASSES];
#pragma unroll
        for (int c = 0; c < N_CLASSES; c++) my_counts[c] = 0;
        for (int r = 0; r < BIN_ROUNDS; r++) {
            const uint32_t sid = chunk * BIN_CHUNK + r * BLOCK + threadIdx.x;
            int cls = -1;
            if (sid < b.n_paths) {
                uint32_t pixel, sample;
                sample_coordinates(b, sid, pixel, sample);
                Rng rng{rng_key(b.seed, pixel, sample), 0};
                V3 org, dir;
                camera_ray(sc, pixel, rng, org, dir);
                surface_plane(res, 0)[sid] = make_double2(0.0, 0.0);
                surface_plane(res, 1)[sid] = make_double2(0.0, 0.0);
                surface_plane(res, 2)[sid] = make_double2(0.0, 0.0);
                const uint32_t medium = static_cast<uint32_t>(sc.background_medium) & 0xFFFFu;
                const Vertex v = next_vertex<TRIS>(sc, b, org, dir, medium, sid, rng.ctr, st);
                cls = v.cls;
                if (v.cls < 0) { // color = throughput (1) * background_color
                    flush_color(res, sid, sc.background[0], sc.background[1], sc.background[2]);
                    pool.cls[sid] = CLASS_DONE;
                }
                else {
                    st2(pool, PL_RAY0, sid, org.x, org.y);
                    st2(pool, PL_RAY1, sid, org.z, dir.x);
                    st2(pool, PL_RAY2, sid, dir.y, dir.z);
                    st2(pool, PL_THR0, sid, 1.0, 1.0);
                    st2(pool, PL_THR1, sid, 1.0, 0.0);
                    st2(pool, PL_SAV0, sid, 0.0, 0.0);
                    st2(pool, PL_SAV1, sid, 0.0, 0.0);
                    st2(pool, PL_COL, sid, 0.0, 0.0);
                    stu4(pool, PL_HIT, sid, static_cast<uint32_t>(__double2loint(v.t)), static_cast<uint32_t>(__double2hiint(v.t)),
                         0u | (RAY_TRANSMITTED << TYPE_SHIFT) | (medium << 16), sid);
                    stu4(pool, PL_IDS, sid, rng.ctr, static_cast<uint32_t>(v.obj), static_cast<uint32_t>(v.prim), 0u);
                    pool.cls[sid] = static_cast<uint8_t>(v.cls);
                }
            }
#pragma unroll
            for (int c = 0; c < N_CLASSES; c++)
                if ((sc.class_mask >> c) & 1u) my_counts[c] += static_cast<uint32_t>(__popcll(__ballot(cls == c)));
        }
        store_chunk_counts(wave_counts, my_counts, block_counts, count_stride, chunk);
    }
}

// Exclusive scan, in class-major order, of block_counts[class * count_stride + chunk] over the
// chunks of the pool `previous` describes, for every class in class_mask; one workgroup.  Leaves the offsets in place and the
// per-class segment starts, counts and the total in `info`.  Tiles of 4096 counts: one uint4 per
// thread, shuffles inside a wave, 16 wave sums through LDS, a running carry across tiles.
__global__ __launch_bounds__(1024) void k_bin_scan(uint32_t* block_counts, uint32_t count_stride, uint32_t class_mask, uint32_t rounds,
                                                   const BinInfo* previous, BinInfo* info) {
    const uint32_t n_blocks = chunk_table(previous, rounds * BLOCK).first_chunk[N_CLASSES];
    __shared__ uint32_t wave_sums[2][16];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t carry = 0; // the same in every thread
    int parity = 0;
    for (int c = 0; c < N_CLASSES; c++) {
        const uint32_t class_start = carry;
        if ((class_mask >> c) & 1u) {
            uint32_t* row = block_counts + static_cast<size_t>(c) * count_stride;
            for (uint32_t base = 0; base < n_blocks; base += 4096) {
                const uint32_t i = base + threadIdx.x * 4;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (i + 3 < n_blocks) v = *reinterpret_cast<const uint4*>(row + i);
                else {
                    if (i < n_blocks) v.x = row[i];
                    if (i + 1 < n_blocks) v.y = row[i + 1];
                    if (i + 2 < n_blocks) v.z = row[i + 2];
                }
                const uint32_t mine = v.x + v.y + v.z + v.w;
                uint32_t incl = mine;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    uint32_t up = __shfl_up(incl, off);
                    if (lane >= static_cast<uint32_t>(off)) incl += up;
                }
                if (lane == 63) wave_sums[parity][wave] = incl;
                __syncthreads();
                uint32_t before = 0, tile_total = 0;
#pragma unroll
                for (uint32_t w = 0; w < 16; w++) {
                    uint32_t sum = wave_sums[parity][w];
                    if (w < wave) before += sum;
                    tile_total += sum;
                }
                parity ^= 1; // the other buffer is free: every wave passed the barrier after reading it
                uint32_t e = carry + before + incl - mine;
                uint4 o;
                o.x = e;
                o.y = e + v.x;
                o.z = o.y + v.y;
                o.w = o.z + v.z;
                if (i + 3 < n_blocks) *reinterpret_cast<uint4*>(row + i) = o;
                else {
                    if (i < n_blocks) row[i] = o.x;
                    if (i + 1 < n_blocks) row[i + 1] = o.y;
                    if (i + 2 < n_blocks) row[i + 2] = o.z;
                }
                carry += tile_total;
            }
        }
        if (threadIdx.x == 0) {
            info->start[c] = class_start;
            info->count[c] = carry - class_start;
        }
    }
    if (threadIdx.x == 0) info->total = carry;
}

// Writes the class-sorted index list: a record that continues with class c goes to
// block_offsets[c][chunk] + (number of class-c records before it in its chunk).  Records of paths
// that ended are left out.  One block per chunk of the pool `previous` describes.
__global__ __launch_bounds__(BLOCK) void k_bin_scatter(PathPool pool, const BinInfo* previous, const uint32_t* block_offsets,
                                                       uint32_t count_stride, uint32_t class_mask, uint32_t rounds, uint32_t* list_out) {
    const ChunkTable table = chunk_table(previous, rounds * BLOCK);
    if (blockIdx.x >= table.first_chunk[N_CLASSES]) return;
    int chunk_class;
    uint32_t first, n;
    locate_chunk(previous, table, rounds * BLOCK, blockIdx.x, chunk_class, first, n);
    __shared__ uint32_t counts[BIN_ROUNDS][BLOCK / 64][N_CLASSES];
    const int wave = threadIdx.x >> 6;
    int classes[BIN_ROUNDS];
    uint32_t below[BIN_ROUNDS];
#pragma unroll
    for (int r = 0; r < BIN_ROUNDS; r++) {
        const uint32_t i = r * BLOCK + threadIdx.x;
        int cls = -1;
        if (i < n) {
            const uint8_t c = pool.cls[first + i];
            if (c < N_CLASSES) cls = c;
        }
        classes[r] = cls;
        below[r] = 0;
#pragma unroll
        for (int c = 0; c < N_CLASSES; c++) {
            if (!((class_mask >> c) & 1u)) continue;
            const unsigned long long m = __ballot(cls == c);
            if (cls == c) below[r] = lanes_below(m);
            if ((threadIdx.x & 63) == 0) counts[r][wave][c] = static_cast<uint32_t>(__popcll(m));
        }
    }
    __syncthreads();
    // exclusive prefix over (round, wave) for each class, plus the chunk's global offset
    if (threadIdx.x < N_CLASSES && ((class_mask >> threadIdx.x) & 1u)) {
        uint32_t running = block_offsets[threadIdx.x * count_stride + blockIdx.x];
        for (int r = 0; r < BIN_ROUNDS; r++)
            for (int w = 0; w < BLOCK / 64; w++) {
                uint32_t v = counts[r][w][threadIdx.x];
                counts[r][w][threadIdx.x] = running;
                running += v;
            }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < BIN_ROUNDS; r++)
        if (classes[r] >= 0) list_out[counts[r][wave][classes[r]] + below[r]] = first + r * BLOCK + threadIdx.x;
}

struct Record { // the ten planes of one path record
    double2 r0, r1, r2, t0, t1, v0, v1, cc;
    uint4 hit, ids;
};
PTK_DEV Record load_record(const PathPool& p, uint32_t i) {
    Record r;
    r.r0 = ld2(p, PL_RAY0, i);
    r.r1 = ld2(p, PL_RAY1, i);
    r.r2 = ld2(p, PL_RAY2, i);
    r.t0 = ld2(p, PL_THR0, i);
    r.t1 = ld2(p, PL_THR1, i);
    r.v0 = ld2(p, PL_SAV0, i);
    r.v1 = ld2(p, PL_SAV1, i);
    r.cc = ld2(p, PL_COL, i);
    r.hit = ldu4(p, PL_HIT, i);
    r.ids = ldu4(p, PL_IDS, i);
    return r;
}

// -------------------------------------------------------------------------------------------------
// A path vertex while it is being shaded.
struct PathVertex {
    V3 point, dir; // the vertex, and the direction the path arrived along
    double t;      // distance travelled to it
    int obj, prim;
    uint32_t sid, depth, medium;
    V3 thr, saved, color;
    double scatter_pdf;
    uint32_t ray_type;
    bool has_hit;
    Rng rng;
    // results of shading: the new direction and medium, and the connection to a light, if any
    V3 n_dir;
    int next_medium;
    bool connect;
    int light_index;
    V3 sh_dir, sh_contrib;
    double sh_dist;
    uint64_t sh_key;
};

// Loop body of raytrace (app/pathtracer.cpp:69-155) for a vertex of class CLS: emission with MIS,
// next-event sampling up to the visibility march, BRDF sampling, medium switch.
template<int CLS, bool TRIS, bool UV>
PTK_DEV void shade_vertex(const DevScene& sc, const SampleResults& res, const DevMedium& med, PathVertex& v) {
    V3& color = v.color;
    V3& thr = v.thr;
    V3& saved = v.saved;
    double& scatter_pdf = v.scatter_pdf;
    uint32_t& ray_type = v.ray_type;
    bool& has_hit = v.has_hit;
    Rng& rng = v.rng;
    V3& n_dir = v.n_dir;
    int& next_medium = v.next_medium;
    bool& connect = v.connect;
    int& light_index = v.light_index;
    V3& sh_dir = v.sh_dir;
    V3& sh_contrib = v.sh_contrib;
    double& sh_dist = v.sh_dist;
    uint64_t& sh_key = v.sh_key;
    const V3 point = v.point, dir = v.dir;
    const double t = v.t;
    const uint32_t depth = v.depth, medium = v.medium, sid = v.sid;
    if (CLS == CLASS_SCATTER) {
        // :71-93
        color = color + medium_emission(med) * thr;
        thr = thr * medium_sample_weight(med, t, true);
        n_dir = sample_spherical(rng);
        if (sc.nee) {
            LightSample ls;
            if (pick_light_point<TRIS>(sc, -1, point, ls, rng)) {
                double weight = mis_weight(ls.light_pdf, PHASE_ISOTROPIC);
                sh_contrib = thr * (weight * PHASE_ISOTROPIC / ls.light_pdf * static_cast<double>(sc.n_lights));
                sh_dir = ls.direction;
                sh_dist = ls.distance;
                sh_key = rng_fork(rng.key, rng.ctr);
                rng.ctr++;
                light_index = ls.light_index;
                connect = true;
            }
            ray_type = RAY_DIFFUSE;
            scatter_pdf = PHASE_ISOTROPIC;
            saved = point;
        }
    }
    else {
        // :94-155
        thr = thr * medium_sample_weight(med, t, false);
        const int obj = v.obj;
        const int prim = v.prim;
        const DevObject& o = sc.objects[obj];
        const DevMaterial& m = o.mat;
        Surface s;
        s.point = point;
        s.incident = dir;
        s.obj = obj;
        s.prim = prim;
        V3 normal = object_normal<TRIS>(sc, o, point, prim);
        s.outside = dot(dir, normal) < 0;
        s.normal = s.outside ? normal : -normal;
        s.use_uv = UV && m.needs_uv != 0;
        s.u = s.v = 0;
        if (s.use_uv) object_uv<TRIS>(sc, o, point, prim, s.u, s.v);

        if (!has_hit) {
            surface_plane(res, 0)[sid] = make_double2(point.x, point.y);
            surface_plane(res, 1)[sid] = make_double2(point.z, s.normal.x);
            surface_plane(res, 2)[sid] = make_double2(s.normal.y, s.normal.z);
            has_hit = true;
        }
        const bool is_specular = ray_type == RAY_REFLECTED || ray_type == RAY_TRANSMITTED;
        if (m.is_light) { // :106-119
            double weight = 1;
            if (sc.nee && depth != 0 && !is_specular)
                weight = mis_weight(scatter_pdf, object_light_pdf<TRIS>(sc, o, point, saved, prim));
            color = color + material_emittance(sc, m, s.use_uv, s.u, s.v) * weight * thr;
        }
        if (sc.nee) { // sample_light up to the march (src/objects.cpp:515-551, 562-578)
            LightSample ls;
            if (pick_light_point<TRIS>(sc, obj, point, ls, rng)) {
                V3 brdf = material_eval<CLS>(sc, m, s, ls.direction, rng);
                if (len2(brdf) != 0) {
                    double pdf = material_pdf<CLS>(sc, m, s, ls.direction);
                    const uint64_t key = rng_fork(rng.key, rng.ctr);
                    rng.ctr++;
                    bool wrong_side = dot(s.incident, s.normal) * dot(ls.direction, s.normal) > 0;
                    double cosine = fmax(dot(s.normal, ls.direction), 0.0);
                    if (!wrong_side && cosine > 0) {
                        double weight = mis_weight(ls.light_pdf, pdf);
                        sh_contrib = brdf * thr * (weight * cosine / ls.light_pdf * static_cast<double>(sc.n_lights));
                        sh_dir = ls.direction;
                        sh_dist = ls.distance;
                        sh_key = key;
                        light_index = ls.light_index;
                        connect = true;
                    }
                }
            }
        }
        BrdfSample bs = material_sample<CLS>(sc, m, s, rng); // :126-137
        bool is_virtual = allow_direct_light(sc, m, s, rng);
        if (is_virtual) {
            bs.type = ray_type;
        }
        else {
            scatter_pdf = bs.pdf;
            saved = point;
        }
        thr = thr * bs.weight;
        if (dot(s.incident, s.normal) * dot(bs.outgoing, s.normal) > 0) // :139-151
            next_medium = s.outside ? m.internal_medium : m.external_medium;
        n_dir = bs.outgoing;
        ray_type = bs.type;
    }
}

// Record -> vertex, the shading of its class, Russian roulette (app/pathtracer.cpp:69-171 without the
// ray casts).  Returns whether the path continues.
template<bool TRIS, bool UV>
PTK_DEV bool shade_record(const DevScene& sc, const BatchParams& b, const SampleResults& res, const Record& cur, int cls, PathVertex& v) {
    const V3 org = v3(cur.r0.x, cur.r0.y, cur.r1.x);
    v.dir = v3(cur.r1.y, cur.r2.x, cur.r2.y);
    v.thr = v3(cur.t0.x, cur.t0.y, cur.t1.x);
    v.saved = v3(cur.v0.x, cur.v0.y, cur.v1.x);
    v.color = v3(cur.v1.y, cur.cc.x, cur.cc.y);
    v.scatter_pdf = cur.t1.y;
    const uint32_t flags = cur.hit.z;
    v.sid = cur.hit.w;
    v.depth = flags & DEPTH_MASK;
    v.ray_type = (flags >> TYPE_SHIFT) & 3u;
    v.has_hit = (flags & FLAG_HAS_HIT) != 0;
    v.medium = flags >> 16;
    if (v.medium == MEDIUM_NONE) v.medium = static_cast<uint32_t>(sc.background_medium);
    const DevMedium& med = sc.media[v.medium];
    uint32_t pixel, sample;
    sample_coordinates(b, v.sid, pixel, sample);
    v.rng = Rng{rng_key(b.seed, pixel, sample), cur.ids.x};
    if (med.type == PTK_MEDIUM_HOMOGENEOUS_SCATTERING) {
        // the numbers drawn for the free-flight distance when this vertex was found: the channel,
        // and the distance only when that channel attenuates (src/medium.cpp:45-51)
        int channel = v.rng.integer(0, 3);
        if (med.extinction[channel] != 0) v.rng.ctr++;
    }
    v.t = __hiloint2double(static_cast<int>(cur.hit.y), static_cast<int>(cur.hit.x));
    v.point = org + v.dir * v.t;
    v.obj = static_cast<int>(cur.ids.y);
    v.prim = static_cast<int>(cur.ids.z);
    v.next_medium = static_cast<int>(v.medium);
    v.connect = false;
    v.light_index = 0;
    v.sh_dir = v3(0.0);
    v.sh_contrib = v3(0.0);
    v.sh_dist = 0;
    v.sh_key = 0;
    switch (cls) {
        case 0: shade_vertex<0, TRIS, UV>(sc, res, med, v); break;
        case 1: shade_vertex<1, TRIS, UV>(sc, res, med, v); break;
        case 2: shade_vertex<2, TRIS, UV>(sc, res, med, v); break;
        case 3: shade_vertex<3, TRIS, UV>(sc, res, med, v); break;
        case 4: shade_vertex<4, TRIS, UV>(sc, res, med, v); break;
        case 5: shade_vertex<5, TRIS, UV>(sc, res, med, v); break;
        case 6: shade_vertex<6, TRIS, UV>(sc, res, med, v); break;
        default: shade_vertex<CLASS_SCATTER, TRIS, UV>(sc, res, med, v); break;
    }
    // Russian roulette :157-171
    bool allow = true;
    if (static_cast<int>(v.depth) >= sc.min_depth) {
        double threshold = fmin(vmax(v.thr), 0.9);
        allow = v.rng.u01() < threshold;
        if (allow) v.thr = v.thr / threshold;
    }
    return allow && static_cast<int>(v.depth) + 1 <= sc.max_depth;
}

PTK_DEV uint32_t medium_bits_of(const PathVertex& v) {
    return v.next_medium < 0 ? MEDIUM_NONE : static_cast<uint32_t>(v.next_medium);
}

// The two rays that leave a shaded vertex: [0] the shadow ray of the connection, [1] the extension
// ray up to its free-flight distance.  A ray that is not wanted gets t_max = -infinity.
template<bool TRIS>
PTK_DEV void vertex_rays(const PathVertex& v, bool alive, double scatter_distance, RayState (&rays)[2]) {
    rays[0] = ray_state<TRIS>(v.point, v.sh_dir, v.connect ? INF : -INF);
    rays[1] = ray_state<TRIS>(v.point, v.n_dir, alive ? scatter_distance : -INF);
}

// The rest of the vertex once the closest hits of its two rays are known: the visibility march
// (compute_visibility, src/objects.cpp:470-513, and the acceptance test of sample_light :558, on its
// own forked random-number stream), the head of the next loop iteration (app/pathtracer.cpp:53-67:
// where does the new ray end), and the record of the continuing path.  Returns the class of the
// next vertex, -1 when the path ended.
template<bool TRIS, bool UV, bool PASS>
PTK_DEV int finish_vertex(const DevScene& sc, const PathPool& out, const SampleResults& res, uint32_t dst, PathVertex& v, bool alive,
                          double scatter_distance, const HitInfo& shadow_hit, const HitInfo& extension_hit, TraversalStack& st,
                          uint32_t* stat) {
    if (v.connect) {
        Rng march_rng{v.sh_key, 0};
        stat[2]++;
        stat[0]++;
        V3 visible = march_to_light<TRIS, UV, PASS>(sc, v.point, v.sh_dir, static_cast<int>(v.medium), v.light_index, v.sh_dist, march_rng, st,
                                              stat[0], shadow_hit);
        v.color = v.color + v.sh_contrib * visible;
    }
    Vertex nv;
    nv.cls = -1;
    if (alive) {
        stat[1]++;
        nv = vertex_of(extension_hit, scatter_distance);
        if (nv.cls < 0 && !sc.background_is_black) v.color = v.color + v.thr * v3(sc.background);
    }
    if (nv.cls < 0) { // the path ended here
        flush_color(res, v.sid, v.color.x, v.color.y, v.color.z);
        out.cls[dst] = CLASS_DONE;
        return -1;
    }
    stat[3]++;
    st2(out, PL_RAY0, dst, v.point.x, v.point.y);
    st2(out, PL_RAY1, dst, v.point.z, v.n_dir.x);
    st2(out, PL_RAY2, dst, v.n_dir.y, v.n_dir.z);
    st2(out, PL_THR0, dst, v.thr.x, v.thr.y);
    st2(out, PL_THR1, dst, v.thr.z, v.scatter_pdf);
    st2(out, PL_SAV0, dst, v.saved.x, v.saved.y);
    st2(out, PL_SAV1, dst, v.saved.z, v.color.x);
    st2(out, PL_COL, dst, v.color.y, v.color.z);
    stu4(out, PL_HIT, dst, static_cast<uint32_t>(__double2loint(nv.t)), static_cast<uint32_t>(__double2hiint(nv.t)),
         (v.depth + 1) | (v.ray_type << TYPE_SHIFT) | (v.has_hit ? FLAG_HAS_HIT : 0u) | (medium_bits_of(v) << 16), v.sid);
    stu4(out, PL_IDS, dst, v.rng.ctr, static_cast<uint32_t>(nv.obj), static_cast<uint32_t>(nv.prim), 0u);
    out.cls[dst] = static_cast<uint8_t>(nv.cls);
    return nv.cls;
}

// One vertex of every path: the loop body of raytrace (app/pathtracer.cpp:69-171) followed by the
// head of its next iteration (:53-67).  A block works through chunks of the class-sorted list (every
// chunk is of one class, so the class switch is a scalar branch); for chunk records
// [first, first + n) it reads record list[first + j] of `in` and writes record first + j of `out`,
// and leaves the counts of the classes it wrote in block_counts[class][chunk].
// TRIS: the scene has triangles or meshes; UV: some material reads a map larger than 1x1 (texture
// coordinates are needed).  Scenes without them get kernels without that code and its registers.
// DEFER: vertices with a ray that enters the bounds of a mesh are queued for k_traverse and
// k_bounce_resume instead of being finished here.  PASS (DEFER kernels; true elsewhere): some material
// lets connections through; without one these kernels contain no hierarchy walk at all.
// statistics: per block [STATISTICS_SLOTS][N_STATISTICS].
template<bool TRIS, bool UV, bool DEFER, bool PASS>
__global__ __launch_bounds__(BLOCK) PTK_WAVES_PER_SIMD(UV ? PTK_BOUNCE_UV_WAVES : PTK_BOUNCE_WAVES) void k_bounce(DevScene sc, BatchParams b, PathPool in, PathPool out,
                                                                                      SampleResults res, const uint32_t* list,
                                                                                      const BinInfo* info, uint32_t rounds,
                                                                                      uint32_t* block_counts, uint32_t count_stride,
                                                                                      unsigned long long* statistics, DeferQueues queues) {
    // DEFER: only the later segments of a visibility march walk a hierarchy here, and only if PASS
    __shared__ StackEntry lds_stack[TRIS && (!DEFER || PASS) ? LDS_STACK_DEPTH * BLOCK : 1];
    __shared__ uint32_t wave_sums[BLOCK / 64][N_STATISTICS];
    __shared__ uint32_t wave_counts[BLOCK / 64][N_CLASSES];
    __shared__ uint32_t queue_counts[BLOCK / 64][2];
    __shared__ uint32_t queue_base[2];
    TraversalStack st = make_stack(lds_stack);
    const ChunkTable table = chunk_table(info, rounds * BLOCK);
#ifdef PTK_STAMPS
#define PTK_STAMP(k)                                                                                   \
    {                                                                                                  \
        const uint64_t now_ = __builtin_readcyclecounter();                                            \
        if ((threadIdx.x & 63) == 0) stat[6 + (k)] += static_cast<uint32_t>((now_ - stamp_) >> 4);     \
        stamp_ = now_;                                                                                 \
    }
    uint64_t stamp_ = __builtin_readcyclecounter();
#else
#define PTK_STAMP(k)
#endif
    for (uint32_t chunk = blockIdx.x; chunk < table.first_chunk[N_CLASSES]; chunk += gridDim.x) {
        int cls;
        uint32_t first, n;
        locate_chunk(info, table, rounds * BLOCK, chunk, cls, first, n);
        uint32_t stat[N_STATISTICS] = {};
        uint32_t my_counts[N_CLASSES];
#pragma unroll
        for (int c = 0; c < N_CLASSES; c++) my_counts[c] = 0;
        // Software pipeline over the rounds of the chunk: the record of the next round is requested
        // in the middle of the current one (after the table lookups of shading, before the two ray
        // casts), and its list entry one round earlier still, so neither latency is paid at the top.
        uint32_t qi = threadIdx.x;
        Record cur{};
        uint32_t src = qi < n ? list[first + qi] : 0u;
        if (qi < n) cur = load_record(in, src);
        uint32_t src_next = qi + BLOCK < n ? list[first + qi + BLOCK] : 0u;
        for (uint32_t r = 0; r * BLOCK < n; r++, qi += BLOCK) {
            const bool have = qi < n, have_next = qi + BLOCK < n;
            const uint32_t dst = first + qi;
            int new_cls = -1;
            Record nxt{};
            uint32_t src_after = 0;
            RayState rays[2];
            bool deferred = false;
            V3 origin = v3(0.0);
            if (have) {
                PathVertex v;
                const bool alive = shade_record<TRIS, UV>(sc, b, res, cur, cls, v);
                PTK_STAMP(0)

                // request the next round's record and the list entry of the one after
                if (have_next) nxt = load_record(in, src_next);
                if (qi + 2 * BLOCK < n) src_after = list[first + qi + 2 * BLOCK];

                // the shadow ray of the connection and the extension ray leave the vertex together
                double scatter_distance = INF;
                if (alive) scatter_distance = free_flight_distance(sc, b, medium_bits_of(v), v.sid, v.rng.ctr);
                vertex_rays<TRIS>(v, alive, scatter_distance, rays);
                scan_objects<TRIS, DEFER ? MESH_DEFER : MESH_INLINE, 2>(sc, v.point, rays, st);
                deferred = DEFER && (rays[0].needs_mesh || rays[1].needs_mesh);
                origin = v.point;
                if (!deferred) new_cls = finish_vertex<TRIS, UV, PASS>(sc, out, res, dst, v, alive, scatter_distance, rays[0].hit, rays[1].hit, st, stat);
                PTK_STAMP(1)
            }
            if (DEFER) { // queue the deferred vertices of this round and their rays: two atomics per block
                const uint32_t my_rays = deferred ? (rays[0].needs_mesh ? 1u : 0u) + (rays[1].needs_mesh ? 1u : 0u) : 0u;
                const unsigned long long vertex_mask = __ballot(deferred);
                uint32_t rays_before = my_rays; // inclusive scan over the wave
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    uint32_t up = __shfl_up(rays_before, off);
                    if ((threadIdx.x & 63) >= static_cast<uint32_t>(off)) rays_before += up;
                }
                const int wave = threadIdx.x >> 6;
                if ((threadIdx.x & 63) == 63) {
                    queue_counts[wave][0] = rays_before;
                    queue_counts[wave][1] = static_cast<uint32_t>(__popcll(vertex_mask));
                }
                __syncthreads();
                if (threadIdx.x < 2) {
                    uint32_t total = 0;
                    for (int w = 0; w < BLOCK / 64; w++) total += queue_counts[w][threadIdx.x];
                    queue_base[threadIdx.x] = total ? atomicAdd(queues.counters + threadIdx.x, total) : 0u;
                }
                __syncthreads();
                if (deferred) {
                    uint32_t ray_slot = queue_base[0] + rays_before - my_rays, vertex_slot = queue_base[1] + lanes_below(vertex_mask);
                    for (int w = 0; w < wave; w++) {
                        ray_slot += queue_counts[w][0];
                        vertex_slot += queue_counts[w][1];
                    }
                    queues.vertices[vertex_slot] = DeferredVertex{dst, src, chunk, static_cast<uint32_t>(cls)};
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        if (rays[k].needs_mesh) {
                            // t_max is the closest hit among the other objects: the walk starts culled
                            queues.rays[ray_slot++] = RayRequest{origin.x, origin.y, origin.z, rays[k].dir.x, rays[k].dir.y, rays[k].dir.z,
                                                                 rays[k].limit, 2 * dst + k, 0u};
                        }
                        else queues.results[2 * dst + k] = RayResult{INF, -1, 0};
                    }
                }
                __syncthreads(); // queue_counts and queue_base are rewritten next round
            }
#pragma unroll
            for (int c = 0; c < N_CLASSES; c++)
                if ((sc.class_mask >> c) & 1u) my_counts[c] += static_cast<uint32_t>(__popcll(__ballot(new_cls == c)));
            cur = nxt;
            src = src_next;
            src_next = src_after;
        }
        store_chunk_counts(wave_counts, my_counts, block_counts, count_stride, chunk);
        add_block_statistics(statistics, wave_sums, stat);
    }
}

// The last bounces of a batch, when few paths are left: one lane follows one path to its end,
// without the class sort between bounces (the class switch is per lane here, which costs nothing
// at this size), so the tail of the path-length distribution costs one launch instead of three per
// bounce.  Same per-vertex code as k_bounce, hence the same results.  Round 0 reads record list[i] of
// `in` and writes record i of `out`; later rounds update record i of `out` in place.
template<bool TRIS, bool UV>
__global__ __launch_bounds__(BLOCK) PTK_WAVES_PER_SIMD(UV ? PTK_BOUNCE_UV_WAVES : PTK_BOUNCE_WAVES) void k_tail(DevScene sc, BatchParams b, PathPool in, PathPool out,
                                                                                    SampleResults res, const uint32_t* list,
                                                                                    const BinInfo* info, unsigned long long* statistics) {
    __shared__ StackEntry lds_stack[TRIS ? LDS_STACK_DEPTH * BLOCK : 1];
    __shared__ uint32_t wave_sums[BLOCK / 64][N_STATISTICS];
    TraversalStack st = make_stack(lds_stack);
    const uint32_t n = info->total;
    if (blockIdx.x * BLOCK >= n) return;
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    uint32_t stat[N_STATISTICS] = {};
    if (i < n) {
        const uint32_t first_src = list[i];
        int cls = in.cls[first_src];
        for (int round = 0; cls >= 0; round++) {
            const Record cur = round == 0 ? load_record(in, first_src) : load_record(out, i);
            PathVertex v;
            const bool alive = shade_record<TRIS, UV>(sc, b, res, cur, cls, v);
            if (round > 0) stat[4]++;
            double scatter_distance = INF;
            if (alive) scatter_distance = free_flight_distance(sc, b, medium_bits_of(v), v.sid, v.rng.ctr);
            RayState rays[2];
            vertex_rays<TRIS>(v, alive, scatter_distance, rays);
            scan_objects<TRIS, MESH_INLINE, 2>(sc, v.point, rays, st);
            cls = finish_vertex<TRIS, UV, true>(sc, out, res, i, v, alive, scatter_distance, rays[0].hit, rays[1].hit, st, stat);
            __threadfence(); // the record written above is read back by the next round
        }
    }
    add_block_statistics(statistics, wave_sums, stat);
}

// Closest triangle for every queued ray, one ray per lane: the meshes of the scene in object order,
// each hierarchy walked with the shrinking t_max.
__global__ __launch_bounds__(BLOCK) PTK_WAVES_PER_SIMD(PTK_TRAVERSE_WAVES) void k_traverse(DevScene sc, DeferQueues queues,
                                                                                           unsigned long long* statistics) {
    __shared__ StackEntry lds_stack[LDS_STACK_DEPTH * BLOCK];
    __shared__ uint32_t wave_sums[BLOCK / 64][N_STATISTICS];
    TraversalStack st = make_stack(lds_stack);
    const uint32_t n = queues.counters[0];
    if (blockIdx.x * BLOCK >= n) return;
    const PTK_CONSTANT DevShape* shapes = (const PTK_CONSTANT DevShape*) sc.shapes;
    uint32_t stat[N_STATISTICS] = {}; // [0] rays, [1] nodes visited, [2] triangles tested
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const RayRequest q = queues.rays[i];
        const V3 org = v3(q.ox, q.oy, q.oz), dir = v3(q.dx, q.dy, q.dz);
        const RayPrep rp = prepare_ray(dir);
        RayResult best{q.limit, -1, 0};
        stat[0]++;
        for (int o = 0; o < sc.n_objects; o++) {
            if (shapes[o].type != PTK_OBJ_MESH) continue;
            int prim = -1;
            // A shadow ray (even slot) is blocked by any triangle of a mesh that lets no connection
            // through: the march ends in black whichever triangle it meets first.
            const bool any_hit = (q.slot & 1u) == 0 && shapes[o].f[6] != 0;
            DevShape shape;
            load_shape(shapes + o, shape);
            mesh_closest_hit(sc, shape.bvh_root, shape.first_tri, org, dir, rp, best.t, prim, st, mesh_bound_abs(shape), any_hit);
            if (prim >= 0) {
                best.obj = o;
                best.prim = prim;
                if (any_hit) break;
            }
        }
        queues.results[q.slot] = best;
    }
    stat[1] = st.nodes_visited;
    stat[2] = st.triangles_tested;
    add_block_statistics(statistics, wave_sums, stat);
}

// Finishes the deferred vertices: shades them again from the same records (same random numbers, so
// the same two rays), scans the objects other than meshes, takes the closer of that hit and the
// triangle k_traverse found, and goes on as k_bounce does.  The class counts go to the chunk the
// vertex came from, by atomics (they are spread over as many addresses as there are chunks).
template<bool UV, bool PASS>
__global__ __launch_bounds__(BLOCK) PTK_WAVES_PER_SIMD(UV ? PTK_BOUNCE_UV_WAVES : PTK_BOUNCE_WAVES) void k_bounce_resume(DevScene sc, BatchParams b, PathPool in,
                                                                                             PathPool out, SampleResults res,
                                                                                             uint32_t* block_counts, uint32_t count_stride,
                                                                                             unsigned long long* statistics,
                                                                                             DeferQueues queues) {
    __shared__ StackEntry lds_stack[PASS ? LDS_STACK_DEPTH * BLOCK : 1];
    __shared__ uint32_t wave_sums[BLOCK / 64][N_STATISTICS];
    TraversalStack st = make_stack(lds_stack);
    const uint32_t n = queues.counters[1];
    if (blockIdx.x * BLOCK >= n) return;
    const PTK_CONSTANT DevShape* shapes = (const PTK_CONSTANT DevShape*) sc.shapes;
    uint32_t stat[N_STATISTICS] = {};
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const DeferredVertex e = queues.vertices[i];
        const Record cur = load_record(in, e.src);
        PathVertex v;
        const bool alive = shade_record<true, UV>(sc, b, res, cur, static_cast<int>(e.cls), v);
        double scatter_distance = INF;
        if (alive) scatter_distance = free_flight_distance(sc, b, medium_bits_of(v), v.sid, v.rng.ctr);
        RayState rays[2];
        vertex_rays<true>(v, alive, scatter_distance, rays);
        scan_objects<true, MESH_SKIP, 2>(sc, v.point, rays, st);
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const RayResult found = queues.results[2 * e.dst + k];
            if (found.obj >= 0 && found.t < rays[k].hit.t) {
                rays[k].hit.t = found.t;
                rays[k].hit.obj = found.obj;
                rays[k].hit.prim = found.prim;
                rays[k].hit.cls = shapes[found.obj].cls;
            }
        }
        const int new_cls = finish_vertex<true, UV, PASS>(sc, out, res, e.dst, v, alive, scatter_distance, rays[0].hit, rays[1].hit, st, stat);
        if (new_cls >= 0) atomicAdd(block_counts + static_cast<size_t>(new_cls) * count_stride + e.chunk, 1u);
    }
    add_block_statistics(statistics, wave_sums, stat);
}

// Sums the per-block statistics slots into totals[N_STATISTICS]; one block.
__global__ __launch_bounds__(BLOCK) void k_sum_statistics(const unsigned long long* slots, unsigned long long* totals) {
    __shared__ unsigned long long partial[BLOCK];
    for (int k = 0; k < N_STATISTICS; k++) {
        unsigned long long sum = 0;
        for (int s = threadIdx.x; s < STATISTICS_SLOTS; s += BLOCK) sum += slots[static_cast<size_t>(s) * N_STATISTICS + k];
        partial[threadIdx.x] = sum;
        __syncthreads();
        for (int off = BLOCK / 2; off > 0; off >>= 1) {
            if (threadIdx.x < off) partial[threadIdx.x] += partial[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) totals[k] = partial[0];
        __syncthreads();
    }
}

// -------------------------------------------------------------------------------------------------
// compute_pixel_color's mean over the batch's samples (app/pathtracer.cpp:192-199) and
// raytrace_section's running mean (:232-235, 253-267); one lane per pixel, samples in order.
__global__ __launch_bounds__(BLOCK) void k_accumulate(BatchParams b, SampleResults res, double* image, double* position,
                                                      double* normal, unsigned long long* pixel_counters) {
    for (uint32_t p = blockIdx.x * BLOCK + threadIdx.x; p < b.n_pixels; p += gridDim.x * BLOCK) {
        double sum[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (uint32_t s = 0; s < b.n_samples; s++) {
            const size_t sid = static_cast<size_t>(s) * b.n_pixels + p;
            const double2* c = reinterpret_cast<const double2*>(res.color) + 2 * sid;
            const double2 ca = c[0], cb = c[1];
            const double2 s0 = surface_plane(res, 0)[sid], s1 = surface_plane(res, 1)[sid], s2 = surface_plane(res, 2)[sid];
            sum[0] += ca.x;
            sum[1] += ca.y;
            sum[2] += cb.x;
            sum[3] += s0.x;
            sum[4] += s0.y;
            sum[5] += s1.x;
            sum[6] += s1.y;
            sum[7] += s2.x;
            sum[8] += s2.y;
        }
        const double n_add = static_cast<double>(b.n_samples);
        const size_t idx = static_cast<size_t>(b.start_idx) + p;
        const double n_prev = static_cast<double>(pixel_counters[idx]);
        for (int j = 0; j < 3; j++) {
            image[3 * idx + j] = (image[3 * idx + j] * n_prev + (sum[j] / n_add) * n_add) / (n_prev + n_add);
            position[3 * idx + j] = (position[3 * idx + j] * n_prev + (sum[3 + j] / n_add) * n_add) / (n_prev + n_add);
            normal[3 * idx + j] = (normal[3 * idx + j] * n_prev + (sum[6 + j] / n_add) * n_add) / (n_prev + n_add);
        }
        pixel_counters[idx] += b.n_samples;
    }
}

// Copies per-sample results to [pixel][sample][3] host-order arrays (ptk_trace_samples).
__global__ __launch_bounds__(BLOCK) void k_gather_samples(BatchParams b, SampleResults res, size_t pixel_base,
                                                          double* color, double* position, double* normal) {
    for (uint32_t sid = blockIdx.x * BLOCK + threadIdx.x; sid < b.n_paths; sid += gridDim.x * BLOCK) {
        uint32_t s = sid / b.n_pixels, p = sid - s * b.n_pixels;
        size_t o = 3 * ((pixel_base + p) * b.n_samples + s);
        if (color) {
            const double2* c = reinterpret_cast<const double2*>(res.color) + 2 * static_cast<size_t>(sid);
            const double2 ca = c[0], cb = c[1];
            color[o] = ca.x;
            color[o + 1] = ca.y;
            color[o + 2] = cb.x;
        }
        const double2 s0 = surface_plane(res, 0)[sid], s1 = surface_plane(res, 1)[sid], s2 = surface_plane(res, 2)[sid];
        if (position) {
            position[o] = s0.x;
            position[o + 1] = s0.y;
            position[o + 2] = s1.x;
        }
        if (normal) {
            normal[o] = s1.y;
            normal[o + 1] = s2.x;
            normal[o + 2] = s2.y;
        }
    }
}

// One primary ray per pixel: Camera::make_ray + find_closest_hit with all Hit attributes
// (src/objects.cpp:434-440).
__global__ __launch_bounds__(BLOCK) void k_primary(DevScene sc, BatchParams b, int32_t* object_index, int32_t* primitive_index,
                                                   double* distance, double* position, double* normal) {
    __shared__ StackEntry lds_stack[LDS_STACK_DEPTH * BLOCK];
    TraversalStack st = make_stack(lds_stack);
    for (uint32_t p = blockIdx.x * BLOCK + threadIdx.x; p < b.n_pixels; p += gridDim.x * BLOCK) {
        const uint32_t pixel = static_cast<uint32_t>(b.start_idx) + p;
        Rng rng{rng_key(b.seed, pixel, b.sample_offset), 0};
        V3 org, dir;
        camera_ray(sc, pixel, rng, org, dir);
        HitInfo h;
        bool found = closest_hit<true>(sc, org, dir, INF, h, st);
        V3 point = v3(0.0), n = v3(0.0);
        int prim = 0;
        if (found) {
            const DevObject& o = sc.objects[h.obj];
            point = org + dir * h.t;
            n = object_normal<true>(sc, o, point, h.prim);
            if (!(dot(dir, n) < 0)) n = -n;
            if (o.type == PTK_OBJ_MESH) prim = sc.tri_cold[h.prim].face_index;
        }
        object_index[p] = found ? h.obj : -1;
        primitive_index[p] = prim;
        distance[p] = found ? h.t : INF;
        position[3 * p] = point.x;
        position[3 * p + 1] = point.y;
        position[3 * p + 2] = point.z;
        normal[3 * p] = n.x;
        normal[3 * p + 1] = n.y;
        normal[3 * p + 2] = n.z;
    }
}

// -------------------------------------------------------------------------------------------------
// launchers
// -------------------------------------------------------------------------------------------------
static inline uint32_t grid_for(uint32_t n, uint32_t max_blocks) {
    uint32_t g = (n + BLOCK - 1) / BLOCK;
    if (g < 1) g = 1;
    return g < max_blocks ? g : max_blocks;
}

uint32_t bin_blocks(uint32_t n) {
    return n == 0 ? 1u : (n + BIN_CHUNK - 1) / BIN_CHUNK;
}

// chunks of a pool of at most `bound` records (each class may end in a partial one)
static inline uint32_t chunk_blocks(uint32_t bound, uint32_t rounds) {
    return (bound + rounds * BLOCK - 1) / (rounds * BLOCK) + N_CLASSES;
}

uint32_t chunk_rounds(uint32_t bound) {
    const uint32_t r = bound / (BLOCK * 2048u); // aim at 2048 chunks or more: eight per CU
    return r < 1 ? 1 : (r > BIN_ROUNDS ? BIN_ROUNDS : r);
}

uint32_t chunk_capacity(uint32_t records) {
    uint32_t most = 0;
    for (uint32_t r = 1; r <= BIN_ROUNDS; r++) {
        // the largest pool the host chunks with r rounds
        const uint64_t limit = r == BIN_ROUNDS ? records : std::min<uint64_t>(records, static_cast<uint64_t>(r + 1) * BLOCK * 2048u);
        most = std::max(most, chunk_blocks(static_cast<uint32_t>(limit), r));
    }
    return most;
}

void launch_generate(hipStream_t s, const DevScene& sc, const BatchParams& b, const PathPool& pool, const SampleResults& res,
                     BinInfo* first_info, uint32_t* block_counts, uint32_t count_stride) {
    const dim3 grid(min(bin_blocks(b.n_paths), static_cast<uint32_t>(STATISTICS_SLOTS))), block(BLOCK);
    if (sc.has_triangles) hipLaunchKernelGGL(k_generate<true>, grid, block, 0, s, sc, b, pool, res, first_info, block_counts, count_stride);
    else hipLaunchKernelGGL(k_generate<false>, grid, block, 0, s, sc, b, pool, res, first_info, block_counts, count_stride);
}

void launch_bin(hipStream_t s, const DevScene& sc, const PathPool& pool, uint32_t bound, uint32_t rounds, const BinInfo* previous,
                uint32_t* block_counts, uint32_t count_stride, BinInfo* info, uint32_t* list_out) {
    hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(1024), 0, s, block_counts, count_stride, sc.class_mask, rounds, previous, info);
    hipLaunchKernelGGL(k_bin_scatter, dim3(chunk_blocks(bound, rounds)), dim3(BLOCK), 0, s, pool, previous, block_counts, count_stride,
                       sc.class_mask, rounds, list_out);
}

void launch_bounce(hipStream_t s, uint32_t bound, uint32_t rounds, const DevScene& sc, const BatchParams& b, const PathPool& in,
                   const PathPool& out, const SampleResults& res, const uint32_t* list, const BinInfo* info, uint32_t* block_counts,
                   uint32_t count_stride, unsigned long long* statistics, const DeferQueues& queues) {
    const dim3 grid(min(chunk_blocks(bound, rounds), static_cast<uint32_t>(STATISTICS_SLOTS))), block(BLOCK);
#define PTK_LAUNCH_BOUNCE(T, U, D, P)                                                                                                  \
    hipLaunchKernelGGL((k_bounce<T, U, D, P>), grid, block, 0, s, sc, b, in, out, res, list, info, rounds, block_counts, count_stride, \
                       statistics, queues)
    if (sc.has_meshes) {
        (void) hipMemsetAsync(queues.counters, 0, 2 * sizeof(uint32_t), s);
        if (sc.needs_uv) {
            if (sc.has_pass_through) PTK_LAUNCH_BOUNCE(true, true, true, true);
            else PTK_LAUNCH_BOUNCE(true, true, true, false);
        }
        else {
            if (sc.has_pass_through) PTK_LAUNCH_BOUNCE(true, false, true, true);
            else PTK_LAUNCH_BOUNCE(true, false, true, false);
        }
    }
    else if (sc.has_triangles) {
        if (sc.needs_uv) PTK_LAUNCH_BOUNCE(true, true, false, true);
        else PTK_LAUNCH_BOUNCE(true, false, false, true);
    }
    else {
        if (sc.needs_uv) PTK_LAUNCH_BOUNCE(false, true, false, true);
        else PTK_LAUNCH_BOUNCE(false, false, false, true);
    }
#undef PTK_LAUNCH_BOUNCE
}

void launch_tail(hipStream_t s, uint32_t bound, const DevScene& sc, const BatchParams& b, const PathPool& in, const PathPool& out,
                 const SampleResults& res, const uint32_t* list, const BinInfo* info, unsigned long long* statistics) {
    const dim3 grid((bound + BLOCK - 1) / BLOCK + 1), block(BLOCK);
#define PTK_LAUNCH_TAIL(T, U) hipLaunchKernelGGL((k_tail<T, U>), grid, block, 0, s, sc, b, in, out, res, list, info, statistics)
    if (sc.has_triangles) {
        if (sc.needs_uv) PTK_LAUNCH_TAIL(true, true);
        else PTK_LAUNCH_TAIL(true, false);
    }
    else {
        if (sc.needs_uv) PTK_LAUNCH_TAIL(false, true);
        else PTK_LAUNCH_TAIL(false, false);
    }
#undef PTK_LAUNCH_TAIL
}

void launch_traverse(hipStream_t s, uint32_t bound, const DevScene& sc, const DeferQueues& queues, unsigned long long* statistics) {
    const dim3 grid(min((2 * bound + BLOCK - 1) / BLOCK + 1, static_cast<uint32_t>(STATISTICS_SLOTS)));
    hipLaunchKernelGGL(k_traverse, grid, dim3(BLOCK), 0, s, sc, queues, statistics);
}

void launch_bounce_resume(hipStream_t s, uint32_t bound, const DevScene& sc, const BatchParams& b, const PathPool& in, const PathPool& out,
                          const SampleResults& res, uint32_t* block_counts, uint32_t count_stride, unsigned long long* statistics,
                          const DeferQueues& queues) {
    const dim3 grid(min((bound + BLOCK - 1) / BLOCK + 1, static_cast<uint32_t>(STATISTICS_SLOTS))), block(BLOCK);
#define PTK_LAUNCH_RESUME(U, P) \
    hipLaunchKernelGGL((k_bounce_resume<U, P>), grid, block, 0, s, sc, b, in, out, res, block_counts, count_stride, statistics, queues)
    if (sc.needs_uv) {
        if (sc.has_pass_through) PTK_LAUNCH_RESUME(true, true);
        else PTK_LAUNCH_RESUME(true, false);
    }
    else {
        if (sc.has_pass_through) PTK_LAUNCH_RESUME(false, true);
        else PTK_LAUNCH_RESUME(false, false);
    }
#undef PTK_LAUNCH_RESUME
}

void launch_sum_statistics(hipStream_t s, const unsigned long long* slots, unsigned long long* totals) {
    hipLaunchKernelGGL(k_sum_statistics, dim3(1), dim3(BLOCK), 0, s, slots, totals);
}

void launch_accumulate(hipStream_t s, const BatchParams& b, const SampleResults& res, double* image, double* position, double* normal,
                       unsigned long long* pixel_counters) {
    hipLaunchKernelGGL(k_accumulate, dim3(grid_for(b.n_pixels, 4096)), dim3(BLOCK), 0, s, b, res, image, position, normal,
                       pixel_counters);
}

void launch_gather_samples(hipStream_t s, const BatchParams& b, const SampleResults& res, size_t pixel_base, double* color,
                           double* position, double* normal) {
    hipLaunchKernelGGL(k_gather_samples, dim3(grid_for(b.n_paths, 4096)), dim3(BLOCK), 0, s, b, res, pixel_base, color, position,
                       normal);
}

void launch_primary(hipStream_t s, const DevScene& sc, const BatchParams& b, int32_t* object_index, int32_t* primitive_index,
                    double* distance, double* position, double* normal) {
    hipLaunchKernelGGL(k_primary, dim3(grid_for(b.n_pixels, 4096)), dim3(BLOCK), 0, s, sc, b, object_index, primitive_index,
                       distance, position, normal);
}

} // namespace ptk
