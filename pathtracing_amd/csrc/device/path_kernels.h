// Path-tracing kernel templates for gfx950 and the helpers they share (included by kernels.hip and by the
// translation units under variants/, which instantiate the template kernels a few at a time so that they
// compile in parallel).
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
#pragma once
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
// k_bounce_resume without visibility marches through pass-through materials: little arithmetic, many loads
#ifndef PTK_RESUME_WAVES
#define PTK_RESUME_WAVES 4
#endif
#ifndef PTK_TRAVERSE_WAVES
#define PTK_TRAVERSE_WAVES 5
#endif
#ifndef PTK_TRAVERSE_QUAD_WAVES
#define PTK_TRAVERSE_QUAD_WAVES 5
#endif
#define PTK_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n, n)))
static_assert(BLOCK == BIN_BLOCK, "k_extend and the bin kernels share their block geometry");

PTK_DEV TraversalStack make_stack(StackEntry* lds, StackEntry* spill, uint32_t* overflows) {
    TraversalStack st;
    st.overflows = overflows;
    st.lds = (LdsStackEntry*) lds + threadIdx.x;
    st.stride = blockDim.x;
    st.spill = spill;
    st.sp = 0;
    st.nodes_visited = st.triangles_tested = 0;
    return st;
}
// declares the traversal stack `st` of a kernel over its LDS array; `scene` is the kernel's DevScene argument
#define PTK_TRAVERSAL_STACK(st, lds_array, scene) \
    StackEntry st##_spill[SPILL_STACK_DEPTH];     \
    TraversalStack st = make_stack(lds_array, st##_spill, (scene).stack_overflows)

// Linear pixel index (row * W + x) of pixel p of a batch.
PTK_DEV uint32_t batch_pixel(const BatchParams& b, uint32_t p) {
    if (b.band_pixels == 0) return static_cast<uint32_t>(b.start_idx) + p;
    const uint32_t k = b.band_first + p, band = k / b.band_pixels;
    return static_cast<uint32_t>(b.start_idx) + band * b.band_stride + (k - band * b.band_pixels);
}

PTK_DEV void sample_coordinates(const BatchParams& b, uint32_t sid, uint32_t& pixel, uint32_t& sample) {
    uint32_t s = sid / b.n_pixels;
    pixel = batch_pixel(b, sid - s * b.n_pixels);
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
            double r = c.aperture / 2.0 * fsqrt(rng.u01());
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
// Element `plane` of record i lies at base + record_offset(i) + plane * RECORD_PLANE_STRIDE: one uniform
// base, one 32-bit per-lane offset shared by the ten planes, and the plane as an immediate offset of the
// load or store.
PTK_DEV uint32_t record_offset(uint32_t i) {
    return (i / RECORD_TILE) * RECORD_TILE_BYTES + (i % RECORD_TILE) * 16u;
}
PTK_DEV double2 ld2(const PathPool& p, int plane, uint32_t i) {
    return *reinterpret_cast<const double2*>(p.base + record_offset(i) + plane * RECORD_PLANE_STRIDE);
}
PTK_DEV void st2(const PathPool& p, int plane, uint32_t i, double a, double b) {
    *reinterpret_cast<double2*>(p.base + record_offset(i) + plane * RECORD_PLANE_STRIDE) = make_double2(a, b);
}
PTK_DEV uint4 ldu4(const PathPool& p, int plane, uint32_t i) {
    return *reinterpret_cast<const uint4*>(p.base + record_offset(i) + plane * RECORD_PLANE_STRIDE);
}
PTK_DEV void stu4(const PathPool& p, int plane, uint32_t i, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    *reinterpret_cast<uint4*>(p.base + record_offset(i) + plane * RECORD_PLANE_STRIDE) = make_uint4(a, b, c, d);
}

// The path ended: its radiance goes to the per-sample results.
// has_surface: the path shaded a surface, so its first-surface position and normal are in the `surface` planes;
// for the other paths those planes hold stale values that the readers replace by zeros.
PTK_DEV void flush_color(const SampleResults& res, uint32_t sid, double r, double g, double bl, bool has_surface) {
    double2* c = reinterpret_cast<double2*>(res.color) + 2 * static_cast<size_t>(sid);
    c[0] = make_double2(r, g);
    c[1] = make_double2(bl, has_surface ? 1.0 : 0.0);
}
// first-surface position and normal of a sample whose radiance element is (ca, cb): planes (px, py), (pz, nx), (ny, nz)
PTK_DEV void load_surface(const SampleResults& res, size_t sid, const double2& cb, double2& s0, double2& s1, double2& s2);
PTK_DEV double2* surface_plane(const SampleResults& res, int plane) {
    return reinterpret_cast<double2*>(res.surface + static_cast<uint64_t>(plane) * res.plane_doubles);
}
PTK_DEV void load_surface(const SampleResults& res, size_t sid, const double2& cb, double2& s0, double2& s1, double2& s2) {
    s0 = s1 = s2 = make_double2(0.0, 0.0);
    if (cb.y != 0.0) {
        s0 = surface_plane(res, 0)[sid];
        s1 = surface_plane(res, 1)[sid];
        s2 = surface_plane(res, 2)[sid];
    }
}

// __syncthreads for blocks whose waves exchange data through LDS only: the fences order LDS accesses alone, so a wave
// does not wait at the barrier until its stores to memory have been acknowledged (with __syncthreads it does).
PTK_DEV void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Statistics without atomics on shared lines: every block adds its totals to its own slot.
// values: [0] visibility-march segments, [1] extension rays, [2] connections marched,
// [3] records written for the next bounce, [4] vertices shaded by k_tail after its first round,
// [5], [6] and [7] cycle stamps of the sections of k_bounce in the -DPTK_STAMPS build (else unused).
PTK_DEV void add_block_statistics(unsigned long long* slots, uint32_t (*wave_sums)[N_STATISTICS], uint32_t* mine) {
#pragma unroll
    for (int k = 0; k < N_STATISTICS; k++)
        for (int off = 32; off > 0; off >>= 1) mine[k] += __shfl_down(mine[k], off);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < N_STATISTICS; k++) wave_sums[threadIdx.x >> 6][k] = mine[k];
    }
    lds_barrier();
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
// (Vector types, not arrays: an array that a loop indexes before it is unrolled can end up in scratch memory, and the
// table is read in every round.)
typedef uint32_t UInt8 __attribute__((ext_vector_type(8)));
static_assert(N_CLASSES == 8, "ChunkTable holds one vector element per class");
struct ChunkTable {
    UInt8 first_chunk;    // chunks of class c: [first_chunk[c], first_chunk[c + 1]), the last class up to n_chunks
    uint32_t n_chunks;
    UInt8 start, count;   // the class segments of the sorted list (BinInfo), read once
};
PTK_DEV ChunkTable chunk_table(const BinInfo* info, uint32_t chunk_size) {
    ChunkTable t;
    uint32_t running = 0;
#pragma unroll
    for (int c = 0; c < N_CLASSES; c++) {
        // (readfirstlane: the values are the same in every lane, and the compiler is to keep them in scalar registers)
        t.start[c] = __builtin_amdgcn_readfirstlane(info->start[c]);
        t.count[c] = __builtin_amdgcn_readfirstlane(info->count[c]);
        t.first_chunk[c] = running;
        running += (t.count[c] + chunk_size - 1) / chunk_size;
    }
    t.n_chunks = running;
    return t;
}
// class, first record and number of records of a chunk (uniform values, selected without indexing the table)
PTK_DEV void locate_chunk(const ChunkTable& t, uint32_t chunk_size, uint32_t chunk, int& cls, uint32_t& first, uint32_t& count) {
    cls = 0;
    uint32_t start = t.start[0], total = t.count[0], first_chunk = t.first_chunk[0];
#pragma unroll
    for (int c = 1; c < N_CLASSES; c++)
        if (chunk >= t.first_chunk[c]) {
            cls = c;
            start = t.start[c];
            total = t.count[c];
            first_chunk = t.first_chunk[c];
        }
    const uint32_t offset = (chunk - first_chunk) * chunk_size;
    first = start + offset;
    count = min(chunk_size, total - offset);
}

// One round of a block's work: BLOCK records of one chunk, lane t taking record first + r * BLOCK + t of the list.
struct RoundCursor {
    uint32_t chunk, first, n, r; // the chunk, its first list entry and record count, the round within it
    int cls;
    bool valid;
};
PTK_DEV RoundCursor first_round(const ChunkTable& t, uint32_t chunk_size, uint32_t chunk, uint32_t n_chunks) {
    RoundCursor c{chunk, 0u, 0u, 0u, 0, chunk < n_chunks};
    if (c.valid) locate_chunk(t, chunk_size, chunk, c.cls, c.first, c.n);
    return c;
}
// the round a block takes after round c: the next of the same chunk, or the first of the block's next chunk
PTK_DEV RoundCursor next_round(const ChunkTable& t, uint32_t chunk_size, const RoundCursor& c, uint32_t n_chunks) {
    if (!c.valid) return c;
    if ((c.r + 1) * BLOCK < c.n) {
        RoundCursor d = c;
        d.r++;
        return d;
    }
    return first_round(t, chunk_size, c.chunk + gridDim.x, n_chunks);
}
PTK_DEV bool in_round(const RoundCursor& c) {
    return c.valid && c.r * BLOCK + threadIdx.x < c.n;
}
PTK_DEV uint32_t list_entry(const uint32_t* list, const RoundCursor& c) {
    return in_round(c) ? list[c.first + c.r * BLOCK + threadIdx.x] : 0u;
}

// Per-class counts of the records a block wrote for one chunk -> block_counts[class][chunk].
PTK_DEV void store_chunk_counts(uint32_t (*wave_counts)[N_CLASSES], const uint32_t* my_counts, uint32_t* block_counts,
                                uint32_t count_stride, uint32_t chunk) {
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int c = 0; c < N_CLASSES; c++) wave_counts[threadIdx.x >> 6][c] = my_counts[c];
    }
    lds_barrier();
    if (threadIdx.x < N_CLASSES) {
        uint32_t total = 0;
        for (int w = 0; w < BLOCK / 64; w++) total += wave_counts[w][threadIdx.x];
        block_counts[threadIdx.x * count_stride + chunk] = total;
    }
    lds_barrier(); // wave_counts is reused by the block's next chunk
}

// The same without a barrier, for counts that are zero when the kernel starts: every wave adds what it wrote itself,
// lane c the count of class c.  (The waves of a block then never wait for one another between chunks; with a barrier
// per chunk they spent about a sixth of their time waiting for the slowest of the four.)  The counts of a chunk sit
// in places of their own, so the atomics of different chunks do not queue on one address.
PTK_DEV void add_chunk_counts(const uint32_t* my_counts, uint32_t* block_counts, uint32_t count_stride, uint32_t chunk) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t mine = 0;
#pragma unroll
    for (int c = 0; c < N_CLASSES; c++)
        if (lane == static_cast<uint32_t>(c)) mine = my_counts[c];
    if (lane < N_CLASSES && mine != 0) atomicAdd(block_counts + lane * count_stride + chunk, mine);
}

// -------------------------------------------------------------------------------------------------
// Camera samples and their first vertex.  The first pool holds the records in sample order as one
// run (described to the class sort as class 0); chunk j = samples [j * BIN_CHUNK, (j + 1) * BIN_CHUNK).
// DEFER (scenes with meshes): a camera ray that enters the bounds of a mesh is queued for k_traverse like the rays of
// a bounce, its record written with the free-flight distance where the distance to the first vertex belongs;
// k_primary_resume completes it and k_count_classes counts the classes of the chunks.
template<bool TRIS, bool DEFER>
__global__ __launch_bounds__(BLOCK) void k_generate(DevScene sc, BatchParams b, PathPool pool, SampleResults res, BinInfo* first_info,
                                                    uint32_t* block_counts, uint32_t count_stride, DeferQueues queues) {
    __shared__ StackEntry lds_stack[TRIS && !DEFER ? LDS_STACK_DEPTH * BLOCK : 1];
    __shared__ uint32_t wave_counts[BLOCK / 64][N_CLASSES];
    __shared__ uint32_t queue_counts[BLOCK / 64];
    __shared__ uint32_t queue_base;
    PTK_TRAVERSAL_STACK(st, lds_stack, sc);
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
            bool deferred = false;
            RayState rays[1];
            V3 org = v3(0.0), dir = v3(0.0);
            if (sid < b.n_paths) {
                uint32_t pixel, sample;
                sample_coordinates(b, sid, pixel, sample);
                Rng rng{rng_key(b.seed, pixel, sample), 0};
                camera_ray(sc, pixel, rng, org, dir);
                const uint32_t medium = static_cast<uint32_t>(sc.background_medium) & 0xFFFFu;
                Vertex v;
                if (DEFER) {
                    const double scatter_distance = free_flight_distance(sc, b, medium, sid, rng.ctr);
                    rays[0] = ray_state<TRIS>(sc, org, dir, scatter_distance);
                    scan_objects<TRIS, MESH_DEFER, 1>(sc, org, rays, st);
                    deferred = rays[0].needs_mesh;
                    v = vertex_of(hit_of(sc, rays[0]), scatter_distance);
                    if (deferred) { // where the ray ends is not known yet
                        v.t = scatter_distance;
                        v.obj = -1;
                        v.prim = 0;
                        v.cls = 0;
                    }
                }
                else v = next_vertex<TRIS>(sc, b, org, dir, medium, sid, rng.ctr, st);
                cls = v.cls;
                if (v.cls < 0) { // color = throughput (1) * background_color
                    flush_color(res, sid, sc.background[0], sc.background[1], sc.background[2], false);
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
                    if (!deferred) pool.cls[sid] = static_cast<uint8_t>(v.cls);
                }
            }
            if (DEFER) { // queue the deferred camera rays of this round: one atomic per block (one ray per vertex)
                const unsigned long long mask = __ballot(deferred);
                const int wave = threadIdx.x >> 6;
                if ((threadIdx.x & 63) == 0) queue_counts[wave] = static_cast<uint32_t>(__popcll(mask));
                lds_barrier();
                if (threadIdx.x == 0) {
                    unsigned long long total = 0;
                    for (int w = 0; w < BLOCK / 64; w++) total += queue_counts[w];
                    const unsigned long long before =
                        total ? atomicAdd(reinterpret_cast<unsigned long long*>(queues.counters), total | (total << 32)) : 0ull;
                    queue_base = static_cast<uint32_t>(before); // rays and vertices advance together
                }
                lds_barrier();
                if (deferred) {
                    uint32_t slot = queue_base + lanes_below(mask);
                    for (int w = 0; w < wave; w++) slot += queue_counts[w];
                    queues.vertices[slot] = sid;
                    queues.rays[slot] = RayRequest{org.x, org.y, org.z, dir.x, dir.y, dir.z, rays[0].limit, 2 * sid + 1, static_cast<uint32_t>(rays[0].obj)};
                }
                lds_barrier(); // queue_counts and queue_base are rewritten next round
            }
            else {
#pragma unroll
                for (int c = 0; c < N_CLASSES; c++)
                    if ((sc.class_mask >> c) & 1u) my_counts[c] += static_cast<uint32_t>(__popcll(__ballot(cls == c)));
            }
        }
        if (!DEFER) store_chunk_counts(wave_counts, my_counts, block_counts, count_stride, chunk);
    }
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

// Record prefetch through LDS: global_load_lds moves 16 bytes per lane from memory straight into LDS
// (lane l of the wave writes dst + 16 l; dst is wave-uniform), without occupying registers while the
// data is on its way.  A wave stages its own 64 records, plane by plane, and reads them back at the
// top of the next round; nothing is shared between waves, so the only synchronisation is the wait for
// the loads before the staging array is read.
typedef __attribute__((address_space(3))) void* LdsVoidPtr;
typedef const __attribute__((address_space(1))) void* GlobalVoidPtr;
PTK_DEV void request_record(const PathPool& p, uint32_t i, double2* wave_stage) {
    __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0): the reads of the record staged before have returned
#pragma unroll
    for (int plane = 0; plane < N_PLANES; plane++)
        __builtin_amdgcn_global_load_lds((GlobalVoidPtr) (p.base + record_offset(i) + plane * RECORD_PLANE_STRIDE), (LdsVoidPtr) (wave_stage + plane * BLOCK), 16, 0, 0);
}
// Every load this wave has requested, the ones into LDS included, has arrived (vmcnt(0); expcnt and lgkmcnt at their
// maxima, i.e. not waited for).  A wave waits for its staged record here, *before* it issues the stores of the round
// the request was made in, and not where it reads the record (the top of the next round): the memory counter
// retires in issue order, so a wait at the top would also wait until every store of the round has been
// acknowledged, which takes microseconds; waves spent a fifth of their lives parked there.
PTK_DEV void wait_for_loads() {
    __builtin_amdgcn_s_waitcnt(0x0F70);
}
// The record request_record asked for, after wait_for_loads.
PTK_DEV Record staged_record(const double2* stage) {
    const double2* mine = stage + threadIdx.x;
    Record r;
    r.r0 = mine[PL_RAY0 * BLOCK];
    r.r1 = mine[PL_RAY1 * BLOCK];
    r.r2 = mine[PL_RAY2 * BLOCK];
    r.t0 = mine[PL_THR0 * BLOCK];
    r.t1 = mine[PL_THR1 * BLOCK];
    r.v0 = mine[PL_SAV0 * BLOCK];
    r.v1 = mine[PL_SAV1 * BLOCK];
    r.cc = mine[PL_COL * BLOCK];
    r.hit = *reinterpret_cast<const uint4*>(mine + PL_HIT * BLOCK);
    r.ids = *reinterpret_cast<const uint4*>(mine + PL_IDS * BLOCK);
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
                sh_contrib = thr * (fdiv(weight * PHASE_ISOTROPIC, ls.light_pdf) * static_cast<double>(sc.n_lights));
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
                        sh_contrib = brdf * thr * (fdiv(weight * cosine, ls.light_pdf) * static_cast<double>(sc.n_lights));
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
    v.point = mad(v.dir, v.t, org);
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
        if (allow) { // (one reciprocal for the three quotients)
            const double r = frcp(threshold);
            v.thr = v3(fdiv_by(v.thr.x, threshold, r), fdiv_by(v.thr.y, threshold, r), fdiv_by(v.thr.z, threshold, r));
        }
    }
    return allow && static_cast<int>(v.depth) + 1 <= sc.max_depth;
}

PTK_DEV uint32_t medium_bits_of(const PathVertex& v) {
    return v.next_medium < 0 ? MEDIUM_NONE : static_cast<uint32_t>(v.next_medium);
}

// The two rays that leave a shaded vertex: [0] the shadow ray of the connection, [1] the extension
// ray up to its free-flight distance.  A ray that is not wanted gets t_max = -infinity.
template<bool TRIS>
PTK_DEV void vertex_rays(const DevScene& sc, const PathVertex& v, bool alive, double scatter_distance, RayState (&rays)[2]) {
    rays[0] = ray_state<TRIS>(sc, v.point, v.sh_dir, v.connect ? INF : -INF);
    rays[1] = ray_state<TRIS>(sc, v.point, v.n_dir, alive ? scatter_distance : -INF);
}

// The rest of the vertex once the closest hits of its two rays are known: the visibility march
// (compute_visibility, src/objects.cpp:470-513, and the acceptance test of sample_light :558, on its
// own forked random-number stream), the head of the next loop iteration (app/pathtracer.cpp:53-67:
// where does the new ray end), and the record of the continuing path.  Returns the class of the
// next vertex, -1 when the path ended.
template<bool TRIS, bool UV, bool PASS>
PTK_DEV Vertex close_vertex(const DevScene& sc, PathVertex& v, bool alive, double scatter_distance, const HitInfo& shadow_hit,
                            const HitInfo& extension_hit, TraversalStack& st, uint32_t* stat) {
    if (v.connect) {
        Rng march_rng{v.sh_key, 0};
        stat[2]++;
        stat[0]++;
        V3 visible = march_to_light<TRIS, UV, PASS>(sc, v.point, v.sh_dir, static_cast<int>(v.medium), v.light_index, v.sh_dist, march_rng, st,
                                              stat[0], shadow_hit);
        v.color = v.color + v.sh_contrib * visible;
    }
    Vertex nv{0.0, -1, 0, -1};
    if (alive) {
        stat[1]++;
        nv = vertex_of(extension_hit, scatter_distance);
        if (nv.cls < 0 && !sc.background_is_black) v.color = v.color + v.thr * v3(sc.background);
    }
    return nv;
}

// Plane PL_HIT of the record of the path that continues from vertex v to the vertex nv.
PTK_DEV void store_hit_plane(const PathPool& out, uint32_t dst, const PathVertex& v, double t) {
    stu4(out, PL_HIT, dst, static_cast<uint32_t>(__double2loint(t)), static_cast<uint32_t>(__double2hiint(t)),
         (v.depth + 1) | (v.ray_type << TYPE_SHIFT) | (v.has_hit ? FLAG_HAS_HIT : 0u) | (medium_bits_of(v) << 16), v.sid);
}

// The record of the path that continues from v; nv: the vertex its extension ray ends at.
PTK_DEV void store_record(const PathPool& out, uint32_t dst, const PathVertex& v, const Vertex& nv) {
    st2(out, PL_RAY0, dst, v.point.x, v.point.y);
    st2(out, PL_RAY1, dst, v.point.z, v.n_dir.x);
    st2(out, PL_RAY2, dst, v.n_dir.y, v.n_dir.z);
    st2(out, PL_THR0, dst, v.thr.x, v.thr.y);
    st2(out, PL_THR1, dst, v.thr.z, v.scatter_pdf);
    st2(out, PL_SAV0, dst, v.saved.x, v.saved.y);
    st2(out, PL_SAV1, dst, v.saved.z, v.color.x);
    st2(out, PL_COL, dst, v.color.y, v.color.z);
    store_hit_plane(out, dst, v, nv.t);
    stu4(out, PL_IDS, dst, v.rng.ctr, static_cast<uint32_t>(nv.obj), static_cast<uint32_t>(nv.prim), 0u);
}

template<bool TRIS, bool UV, bool PASS>
PTK_DEV int finish_vertex(const DevScene& sc, const PathPool& out, const SampleResults& res, uint32_t dst, PathVertex& v, bool alive,
                          double scatter_distance, const HitInfo& shadow_hit, const HitInfo& extension_hit, TraversalStack& st,
                          uint32_t* stat) {
    const Vertex nv = close_vertex<TRIS, UV, PASS>(sc, v, alive, scatter_distance, shadow_hit, extension_hit, st, stat);
    if (nv.cls < 0) { // the path ended here
        flush_color(res, v.sid, v.color.x, v.color.y, v.color.z, v.has_hit);
        out.cls[dst] = CLASS_DONE;
        return -1;
    }
    stat[3]++;
    store_record(out, dst, v, nv);
    out.cls[dst] = static_cast<uint8_t>(nv.cls);
    return nv.cls;
}

// A vertex whose rays wait for k_traverse: its record as far as it is known, and what k_bounce_resume needs beside it.
PTK_DEV void store_pending(const PathPool& out, const DeferQueues& queues, uint32_t dst, const PathVertex& v, bool alive,
                           double scatter_distance) {
    if (alive) {
        Vertex unknown{0.0, -1, 0, -1};
        store_record(out, dst, v, unknown);
    }
    else { // the path ends in k_bounce_resume, which needs where the vertex is, the light gathered so far and the sample index
        st2(out, PL_RAY0, dst, v.point.x, v.point.y);
        st2(out, PL_RAY1, dst, v.point.z, v.n_dir.x);
        st2(out, PL_SAV1, dst, v.saved.z, v.color.x);
        st2(out, PL_COL, dst, v.color.y, v.color.z);
        store_hit_plane(out, dst, v, 0.0);
    }
    double2* planes = reinterpret_cast<double2*>(queues.pending);
    const uint64_t n = queues.capacity;
    if (v.connect) {
        planes[dst] = make_double2(v.sh_dir.x, v.sh_dir.y);
        planes[n + dst] = make_double2(v.sh_dir.z, v.sh_contrib.x);
        planes[2 * n + dst] = make_double2(v.sh_contrib.y, v.sh_contrib.z);
    }
    planes[3 * n + dst] = make_double2(v.sh_dist, scatter_distance);
    const uint32_t flags = (v.medium & 0xFFFFu) | (v.connect ? PENDING_CONNECT : 0u) | (alive ? PENDING_ALIVE : 0u);
    planes[4 * n + dst] = make_double2(__longlong_as_double(static_cast<long long>(v.sh_key)),
                                       __hiloint2double(static_cast<int>(flags), v.light_index));
}

// One vertex of every path: the loop body of raytrace (app/pathtracer.cpp:69-171) followed by the
// head of its next iteration (:53-67).  A block works through chunks of the class-sorted list (every
// chunk is of one class, so the class switch is a scalar branch); for chunk records
// [first, first + n) it reads record list[first + j] of `in` and writes record first + j of `out`,
// and leaves the counts of the classes it wrote in block_counts[class][chunk].
// TRIS: the scene has triangles or meshes; UV: some material reads a map larger than 1x1 (texture
// coordinates are needed).  Scenes without them get kernels without that code and its registers.
// DEFER: vertices with a ray that enters the bounds of a mesh are queued for k_traverse and
// k_bounce_resume instead of being finished here.  PASS: some material lets connections to a light
// through; without one a visibility march is its first segment, and the DEFER kernels contain no
// hierarchy walk at all.  CACHE: the scene's tables fit in LDS (stage_scene).
// statistics: per block [STATISTICS_SLOTS][N_STATISTICS].
// The object, light and medium tables of a scene that is small enough are copied to LDS by every block
// (DevScene::tables_fit_lds, LDS_SCENE_DOUBLES): shading reads them with per-lane indices several times per
// vertex, and each of those dependent reads costs a trip to the vector cache otherwise.  The returned scene
// description points at the copies.
template<bool CACHE>
PTK_DEV DevScene stage_scene(const DevScene& sc, double* lds) {
    DevScene out = sc;
    if (CACHE) {
        const uint32_t n_obj = sc.n_objects * (sizeof(DevObject) / 8), n_light = sc.n_lights * (sizeof(DevObject) / 8),
                       n_med = sc.n_media * (sizeof(DevMedium) / 8);
        const double* src_obj = reinterpret_cast<const double*>(sc.objects);
        const double* src_light = reinterpret_cast<const double*>(sc.light_objects);
        const double* src_med = reinterpret_cast<const double*>(sc.media);
        for (uint32_t i = threadIdx.x; i < n_obj; i += blockDim.x) lds[i] = src_obj[i];
        for (uint32_t i = threadIdx.x; i < n_light; i += blockDim.x) lds[n_obj + i] = src_light[i];
        for (uint32_t i = threadIdx.x; i < n_med; i += blockDim.x) lds[n_obj + n_light + i] = src_med[i];
        lds_barrier();
        out.objects = reinterpret_cast<const DevObject*>(lds);
        out.light_objects = reinterpret_cast<const DevObject*>(lds + n_obj);
        out.media = reinterpret_cast<const DevMedium*>(lds + n_obj + n_light);
    }
    return out;
}

template<bool TRIS, bool UV, bool DEFER, bool PASS, bool CACHE>
__global__ __launch_bounds__(BLOCK) PTK_WAVES_PER_SIMD(UV ? PTK_BOUNCE_UV_WAVES : PTK_BOUNCE_WAVES) void k_bounce(DevScene sc_global, BatchParams b, PathPool in, PathPool out,
                                                                                      SampleResults res, const uint32_t* list,
                                                                                      const BinInfo* info, uint32_t rounds,
                                                                                      uint32_t* block_counts, uint32_t count_stride,
                                                                                      unsigned long long* statistics, DeferQueues queues) {
    // DEFER: only the later segments of a visibility march walk a hierarchy here, and only if PASS
    constexpr bool STACK = TRIS && (!DEFER || PASS);
    __shared__ StackEntry lds_stack[STACK ? LDS_STACK_DEPTH * BLOCK : 1];
    // Kernels without a traversal stack have the LDS to stage the next round's records there (STAGE);
    // the others prefetch them into registers.
    constexpr bool STAGE = !STACK;
    __shared__ double2 lds_record[STAGE ? N_PLANES * BLOCK : 1];
    double2* const wave_stage = lds_record + 64 * __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __shared__ uint32_t wave_sums[BLOCK / 64][N_STATISTICS];
    __shared__ uint32_t queue_counts[2][BLOCK / 64][2]; // two sets, used by alternate rounds
    __shared__ uint32_t queue_base[2][2];
    int queue_set = 0;
    PTK_TRAVERSAL_STACK(st, lds_stack, sc_global);
    __shared__ double lds_scene[CACHE ? LDS_SCENE_DOUBLES : 1];
    const DevScene sc = stage_scene<CACHE>(sc_global, lds_scene);
    const ChunkTable table = chunk_table(info, rounds * BLOCK);
#ifdef PTK_STAMPS
#define PTK_STAMP(k)                                                                                   \
    {                                                                                                  \
        const uint64_t now_ = __builtin_readcyclecounter();                                            \
        if ((threadIdx.x & 63) == 0) stat[(k) == 3 ? 4 : (k) == 2 ? 5 : 6 + (k)] += static_cast<uint32_t>((now_ - stamp_) >> 4); \
        stamp_ = now_;                                                                                 \
    }
    uint64_t stamp_ = __builtin_readcyclecounter();
#else
#define PTK_STAMP(k)
#endif
    // A block works through the rounds of its chunks (chunk blockIdx.x, then every gridDim.x-th) as one sequence,
    // software-pipelined across the chunk boundaries: the record of the next round is requested in the middle of the
    // current one (after the table lookups of shading, before the two ray casts) and its list entry one round earlier
    // still, whether that round belongs to the same chunk or opens the block's next one.  Only the block's very first
    // record is waited for with nothing to do.  (When every chunk began with its own dependent pair of loads, list entry
    // then record, waves spent a fifth of their lives at the tops of chunks.)
    const uint32_t n_chunks = table.n_chunks;
    uint32_t stat[N_STATISTICS] = {};
    uint32_t my_counts[N_CLASSES];
#pragma unroll
    for (int c = 0; c < N_CLASSES; c++) my_counts[c] = 0;
    RoundCursor c0 = first_round(table, rounds * BLOCK, blockIdx.x, n_chunks);
    RoundCursor c1 = next_round(table, rounds * BLOCK, c0, n_chunks);
    Record cur{};
    {
        const uint32_t src = list_entry(list, c0);
        if (in_round(c0)) {
            if (STAGE) request_record(in, src, wave_stage);
            else cur = load_record(in, src);
        }
    }
    uint32_t src_next = list_entry(list, c1);
    if (STAGE) wait_for_loads();
    while (c0.valid) {
        const RoundCursor c2 = next_round(table, rounds * BLOCK, c1, n_chunks);
        const bool have = in_round(c0);
        const uint32_t dst = c0.first + c0.r * BLOCK + threadIdx.x;
        int new_cls = -1;
        if (STAGE && have) cur = staged_record(lds_record);
        PTK_STAMP(3)
        // The list entry requested a round ago is consumed here on every path: were it still on its way
        // on some path when its register is reused below, the wait for it would be placed after the
        // record requests of this round and would wait for them as well (loads return in order).
        asm volatile("" : : "v"(src_next));
        Record nxt{};
        RayState rays[2];
        bool deferred = false;
        V3 origin = v3(0.0);
        PathVertex v;
        bool alive = false;
        if (have) alive = shade_record<TRIS, UV>(sc, b, res, cur, c0.cls, v);
        PTK_STAMP(0)
        // Request the next round's record and the list entry of the one after.  Not earlier: registers
        // would be held all through shading, and the compiler makes every read of the scene tables in
        // LDS wait for outstanding loads into LDS; the object scan that follows reads none.  (Lanes beyond the end
        // of this round's chunk take part: the next round may be the first of another chunk.)
        if (in_round(c1)) {
            if (STAGE) request_record(in, src_next, wave_stage);
            else nxt = load_record(in, src_next);
        }
        const uint32_t src_after = list_entry(list, c2);
        double scatter_distance = INF;
        if (have) {
            // the shadow ray of the connection and the extension ray leave the vertex together
            if (alive) scatter_distance = free_flight_distance(sc, b, medium_bits_of(v), v.sid, v.rng.ctr);
            vertex_rays<TRIS>(sc, v, alive, scatter_distance, rays);
            scan_objects<TRIS, DEFER ? MESH_DEFER : MESH_INLINE, 2>(sc, v.point, rays, st);
        }
        if (STAGE) wait_for_loads(); // the next round's record has had the scan to arrive; the stores begin below
        PTK_STAMP(1)
        if (have) {
            deferred = DEFER && (rays[0].needs_mesh || rays[1].needs_mesh);
            origin = v.point;
            if (!deferred) new_cls = finish_vertex<TRIS, UV, PASS>(sc, out, res, dst, v, alive, scatter_distance, hit_of(sc, rays[0]), hit_of(sc, rays[1]), st, stat);
            else store_pending(out, queues, dst, v, alive, scatter_distance);
        }
        PTK_STAMP(2)
        if (DEFER) { // queue the deferred vertices of this round and their rays: one atomic per block
            const uint32_t my_rays = deferred ? (rays[0].needs_mesh ? 1u : 0u) + (rays[1].needs_mesh ? 1u : 0u) : 0u;
            const unsigned long long vertex_mask = __ballot(deferred);
            uint32_t rays_before = my_rays; // inclusive scan over the wave
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                uint32_t up = __shfl_up(rays_before, off);
                if ((threadIdx.x & 63) >= static_cast<uint32_t>(off)) rays_before += up;
            }
            const int wave = threadIdx.x >> 6;
            uint32_t (*counts)[2] = queue_counts[queue_set];
            uint32_t* base = queue_base[queue_set];
            if ((threadIdx.x & 63) == 63) {
                counts[wave][0] = rays_before;
                counts[wave][1] = static_cast<uint32_t>(__popcll(vertex_mask));
            }
            lds_barrier();
            if (threadIdx.x == 0) {
                // both counters with one 64-bit atomic (rays in the low half, which never carries: a batch queues
                // fewer than 2^32 rays): their one address serves about a hundred atomics per microsecond
                unsigned long long total = 0;
                for (int w = 0; w < BLOCK / 64; w++) total += counts[w][0] + (static_cast<unsigned long long>(counts[w][1]) << 32);
                const unsigned long long before = total ? atomicAdd(reinterpret_cast<unsigned long long*>(queues.counters), total) : 0ull;
                base[0] = static_cast<uint32_t>(before);
                base[1] = static_cast<uint32_t>(before >> 32);
            }
            lds_barrier();
            if (deferred) {
                uint32_t ray_slot = base[0] + rays_before - my_rays, vertex_slot = base[1] + lanes_below(vertex_mask);
                for (int w = 0; w < wave; w++) {
                    ray_slot += counts[w][0];
                    vertex_slot += counts[w][1];
                }
                queues.vertices[vertex_slot] = dst;
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    if (rays[k].needs_mesh) {
                        // t_max is the closest hit among the other objects: the walk starts culled
                        queues.rays[ray_slot++] = RayRequest{origin.x, origin.y, origin.z, rays[k].dir.x, rays[k].dir.y, rays[k].dir.z,
                                                             rays[k].limit, 2 * dst + k, static_cast<uint32_t>(rays[k].obj)};
                    }
                    else queues.results[2 * dst + k] = RayResult{rays[k].limit, rays[k].obj, rays[k].prim}; // what the scan found
                }
            }
            // No barrier here: the next round uses the other set, and a wave reaches the round after that only through
            // the next round's two barriers, which every wave passes after its reads of this set.
            queue_set ^= 1;
        }
        if (!DEFER) { // (a deferring kernel writes part of the chunk's records only: k_count_classes counts afterwards)
#pragma unroll
            for (int c = 0; c < N_CLASSES; c++)
                if ((sc.class_mask >> c) & 1u) my_counts[c] += static_cast<uint32_t>(__popcll(__ballot(new_cls == c)));
        }
        if (!STAGE) cur = nxt;
        src_next = src_after;
        if (!DEFER && !(c1.valid && c1.chunk == c0.chunk)) { // the last round of its chunk
            add_chunk_counts(my_counts, block_counts, count_stride, c0.chunk);
#pragma unroll
            for (int c = 0; c < N_CLASSES; c++) my_counts[c] = 0;
        }
        c0 = c1;
        c1 = c2;
    }
    add_block_statistics(statistics, wave_sums, stat);
}

constexpr uint32_t TAIL_REFILL_LANES = 16; // idle lanes at which a wave of k_tail fetches new paths

// The last bounces of a batch, when few paths are left: one lane follows one path to its end,
// without the class sort between bounces (the class switch is per lane here, which costs nothing
// at this size), so the tail of the path-length distribution costs one launch instead of three per
// bounce.  Same per-vertex code as k_bounce, hence the same results.  Round 0 reads record list[i] of
// `in` and writes record i of `out`; later rounds update record i of `out` in place.
template<bool TRIS, bool UV, bool PASS, bool CACHE>
__global__ __launch_bounds__(BLOCK) PTK_WAVES_PER_SIMD(UV ? PTK_BOUNCE_UV_WAVES : PTK_BOUNCE_WAVES) void k_tail(DevScene sc_global, BatchParams b, PathPool in, PathPool out,
                                                                                    SampleResults res, const uint32_t* list,
                                                                                    BinInfo* info, unsigned long long* statistics) {
    __shared__ StackEntry lds_stack[TRIS ? LDS_STACK_DEPTH * BLOCK : 1];
    __shared__ uint32_t wave_sums[BLOCK / 64][N_STATISTICS];
    PTK_TRAVERSAL_STACK(st, lds_stack, sc_global);
    const uint32_t n = info->total;
    __shared__ double lds_scene[CACHE ? LDS_SCENE_DOUBLES : 1];
    const DevScene sc = stage_scene<CACHE>(sc_global, lds_scene);
    uint32_t stat[N_STATISTICS] = {};
    // Path lengths are geometric (Russian roulette): a wave that kept its 64 paths until the longest ended would run
    // three quarters empty.  Lanes whose path has ended take the next one from the list once 16 of them are idle
    // (info->cursor, zeroed by k_bin_scan); the grid fills the device once and every wave leaves when the list is
    // handed out and its own paths have ended.
    bool exhausted = false; // wave-uniform
    uint32_t i = 0, first_src = 0;
    int cls = -1;           // -1: the lane has no path
    bool fresh = false;     // the path's record is still the one of the input pool
    for (;;) {
        const unsigned long long idle = __ballot(cls < 0);
        const uint32_t n_idle = static_cast<uint32_t>(__popcll(idle));
        if (!exhausted && n_idle >= TAIL_REFILL_LANES) {
            // (one atomic per fetch: a wave that reserves a longer stretch of the list and hands it out to its lanes
            // over several fetches was slower here, 1371 against 1394 Mpaths/s on the Cornell scene: the list is short
            // and its end is then held by few waves)
            uint32_t base = 0;
            if (cls < 0 && lanes_below(idle) == 0) base = atomicAdd(&info->cursor, n_idle); // the first idle lane
            base = __shfl(base, __ffsll(static_cast<long long>(idle)) - 1);
            if (cls < 0) {
                const uint32_t mine = base + lanes_below(idle);
                if (mine < n) {
                    i = mine;
                    first_src = list[i];
                    cls = in.cls[first_src];
                    fresh = true;
                }
            }
            exhausted = base + n_idle >= n;
        }
        if (__ballot(cls >= 0) == 0) {
            if (exhausted) break;
            continue; // all 64 lanes are idle: the next pass fetches
        }
        if (cls >= 0) {
            const Record cur = fresh ? load_record(in, first_src) : load_record(out, i);
            PathVertex v;
            const bool alive = shade_record<TRIS, UV>(sc, b, res, cur, cls, v);
            if (!fresh) stat[4]++;
            fresh = false;
            double scatter_distance = INF;
            if (alive) scatter_distance = free_flight_distance(sc, b, medium_bits_of(v), v.sid, v.rng.ctr);
            RayState rays[2];
            vertex_rays<TRIS>(sc, v, alive, scatter_distance, rays);
            scan_objects<TRIS, MESH_INLINE, 2>(sc, v.point, rays, st);
            cls = finish_vertex<TRIS, UV, PASS>(sc, out, res, i, v, alive, scatter_distance, hit_of(sc, rays[0]), hit_of(sc, rays[1]), st, stat);
            __threadfence(); // the record written above is read back by the next round
        }
    }
    add_block_statistics(statistics, wave_sums, stat);
}

// Finishes the deferred vertices: takes the closest hits of their two rays from the results (k_bounce's object scan,
// overridden by k_traverse where a triangle is closer), runs the visibility march of the connection, and completes the
// record k_bounce began (store_pending): where the extension ray ends, and the light the connection carried.
template<bool UV, bool PASS, bool CACHE>
__global__ __launch_bounds__(BLOCK) PTK_WAVES_PER_SIMD(PASS ? PTK_BOUNCE_WAVES : PTK_RESUME_WAVES) void k_bounce_resume(DevScene sc_global, BatchParams b, PathPool out, SampleResults res,
                                                                                              unsigned long long* statistics, DeferQueues queues) {
    __shared__ StackEntry lds_stack[PASS ? LDS_STACK_DEPTH * BLOCK : 1];
    __shared__ uint32_t wave_sums[BLOCK / 64][N_STATISTICS];
    PTK_TRAVERSAL_STACK(st, lds_stack, sc_global);
    const uint32_t n = queues.counters[1];
    if (blockIdx.x * BLOCK >= n) return;
    __shared__ double lds_scene[CACHE ? LDS_SCENE_DOUBLES : 1];
    const DevScene sc = stage_scene<CACHE>(sc_global, lds_scene);
    uint32_t stat[N_STATISTICS] = {};
    const double2* planes = reinterpret_cast<const double2*>(queues.pending);
    const uint64_t cap = queues.capacity;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const uint32_t dst = queues.vertices[i];
        const RayResult found[2] = {queues.results[2 * dst], queues.results[2 * dst + 1]};
        const double2 p3 = planes[3 * cap + dst], p4 = planes[4 * cap + dst];
        const double2 r0 = ld2(out, PL_RAY0, dst), r1 = ld2(out, PL_RAY1, dst), s1 = ld2(out, PL_SAV1, dst), cc = ld2(out, PL_COL, dst);
        const uint4 hit = ldu4(out, PL_HIT, dst);
        const uint32_t flags = static_cast<uint32_t>(__double2hiint(p4.y));
        double2 p0 = make_double2(0.0, 0.0), p1 = p0, p2 = p0;
        if (flags & PENDING_CONNECT) { // planes 0 to 2 describe the connection
            p0 = planes[dst];
            p1 = planes[cap + dst];
            p2 = planes[2 * cap + dst];
        }
        PathVertex v{};
        v.point = v3(r0.x, r0.y, r1.x);
        v.color = v3(s1.y, cc.x, cc.y);
        v.sid = hit.w;
        v.has_hit = (hit.z & FLAG_HAS_HIT) != 0;
        v.medium = flags & 0xFFFFu;
        v.connect = (flags & PENDING_CONNECT) != 0;
        v.light_index = __double2loint(p4.y);
        v.sh_dir = v3(p0.x, p0.y, p1.x);
        v.sh_contrib = v3(p1.y, p2.x, p2.y);
        v.sh_dist = p3.x;
        v.sh_key = static_cast<uint64_t>(__double_as_longlong(p4.x));
        if (!sc.background_is_black) { // the throughput matters only to a path that leaves the scene
            const double2 t0 = ld2(out, PL_THR0, dst), t1 = ld2(out, PL_THR1, dst);
            v.thr = v3(t0.x, t0.y, t1.x);
        }
        const bool alive = (flags & PENDING_ALIVE) != 0;
        const double scatter_distance = p3.y;
        // The closest hits of the two rays are in the results: what k_bounce's object scan found, replaced by
        // k_traverse where a triangle of a mesh is closer (it starts from the scan's distance and object).
        HitInfo hits[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            no_hit(hits[k]);
            if (found[k].obj >= 0) {
                const DevObject& o = sc.objects[found[k].obj];
                hits[k].t = found[k].t;
                hits[k].obj = found[k].obj;
                // a ray that went through k_traverse without finding a closer triangle carries no triangle index:
                // the object is then not a mesh, and a single triangle's index is its object's first_tri
                hits[k].prim = o.type == PTK_OBJ_MESH ? found[k].prim : (o.type == PTK_OBJ_TRIANGLE ? static_cast<int>(o.first_tri) : 0);
                hits[k].cls = o.mat.type;
            }
        }
        const Vertex nv = close_vertex<true, UV, PASS>(sc, v, alive, scatter_distance, hits[0], hits[1], st, stat);
        if (nv.cls < 0) { // the path ended here
            flush_color(res, v.sid, v.color.x, v.color.y, v.color.z, v.has_hit);
            out.cls[dst] = CLASS_DONE;
        }
        else {
            stat[3]++;
            if (v.connect) {
                st2(out, PL_SAV1, dst, s1.x, v.color.x);
                st2(out, PL_COL, dst, v.color.y, v.color.z);
            }
            stu4(out, PL_HIT, dst, static_cast<uint32_t>(__double2loint(nv.t)), static_cast<uint32_t>(__double2hiint(nv.t)), hit.z, hit.w);
            const uint4 ids = ldu4(out, PL_IDS, dst); // (the counter of the random stream stays)
            stu4(out, PL_IDS, dst, ids.x, static_cast<uint32_t>(nv.obj), static_cast<uint32_t>(nv.prim), 0u);
            out.cls[dst] = static_cast<uint8_t>(nv.cls);
        }
    }
    add_block_statistics(statistics, wave_sums, stat);
}

} // namespace ptk
