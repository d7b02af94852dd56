// Kernels that exist once (class sort, traversal, film accumulation, primary rays) and the launchers of all
// kernels; the path-tracing kernel templates live in path_kernels.h.
#include <algorithm>
#include <string>

#include "path_kernels.h"
#include "variants.h"

namespace ptk {

// Exclusive scan, in class-major order, of block_counts[class * count_stride + chunk] over the
// chunks of the pool `previous` describes, for every class in class_mask; one workgroup.  Leaves the offsets in place and the
// per-class segment starts, counts and the total in `info`.  Tiles of 4096 counts: one uint4 per
// thread, shuffles inside a wave, 16 wave sums through LDS, a running carry across tiles; the counts of the next tile
// (of this class or the next) are requested before the current ones are scanned, and the loop is kept small: the
// kernel runs between two large ones and starts with a cold instruction cache.  Records only ever leave the pool: a
// total above the previous one means the counts are corrupt, and the pool is declared empty rather than letting the
// kernels behind this one index with them.
// Inclusive prefix sum over the 64 lanes of a wave by data-parallel-primitive moves (no trip through LDS): shifts by
// 1, 2, 4, 8 inside each row of 16 lanes (lanes without a source add 0), then the last lane of row 0 / 2 broadcast to
// row 1 / 3, then the last lane of row 1 to rows 2 and 3.  Every lane of the wave must be active.
PTK_DEV uint32_t wave_inclusive_sum(uint32_t x) {
    int v = static_cast<int>(x);
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false); // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false); // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false); // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false); // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false); // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false); // row_bcast:31 into rows 2 and 3
    return static_cast<uint32_t>(v);
}

PTK_DEV uint4 counts_at(const uint32_t* row, uint32_t at, uint32_t n) {
    const uint32_t last = n ? (n - 1) & ~3u : 0; // the load stays inside the row (its stride is a multiple of four entries)
    uint4 v = *reinterpret_cast<const uint4*>(row + min(at, last));
    v.x = at < n ? v.x : 0;
    v.y = at + 1 < n ? v.y : 0;
    v.z = at + 2 < n ? v.z : 0;
    v.w = at + 3 < n ? v.w : 0;
    return v;
}

__global__ __launch_bounds__(1024) void k_bin_scan(uint32_t* block_counts, uint32_t count_stride, uint32_t class_mask, uint32_t rounds,
                                                   const BinInfo* previous, BinInfo* info) {
    const uint32_t n_blocks = chunk_table(previous, rounds * BLOCK).n_chunks;
    __shared__ uint32_t wave_sums[2][16];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t carry = 0, class_start = 0; // the same in every thread
    int parity = 0;
    // (class, tile) pairs in order; classes outside the mask, and every class of an empty pool, have no tiles
    const uint32_t tiles = (n_blocks + 4095) / 4096;
    const uint32_t present = tiles ? class_mask & ((1u << N_CLASSES) - 1u) : 0u;
    uint32_t left = present; // classes still to come, the current one included
    uint32_t tile = 0;
    uint4 next = make_uint4(0, 0, 0, 0);
    if (left) next = counts_at(block_counts + static_cast<size_t>(__builtin_ctz(left)) * count_stride, threadIdx.x * 4, n_blocks);
    while (left) {
        const uint32_t c = __builtin_ctz(left);
        const uint32_t i = tile * 4096 + threadIdx.x * 4;
        uint32_t* row = block_counts + static_cast<size_t>(c) * count_stride;
        const uint4 v = next;
        const bool last_tile = tile + 1 == tiles;
        if (last_tile) {
            left &= left - 1;
            tile = 0;
        } else tile++;
        if (left) next = counts_at(block_counts + static_cast<size_t>(__builtin_ctz(left)) * count_stride, tile * 4096 + threadIdx.x * 4, n_blocks);
        const uint32_t mine = v.x + v.y + v.z + v.w;
        const uint32_t incl = wave_inclusive_sum(mine);
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
        else { // the last, partial group; entries past the last chunk stay as they are (zero)
            if (i < n_blocks) row[i] = o.x;
            if (i + 1 < n_blocks) row[i + 1] = o.y;
            if (i + 2 < n_blocks) row[i + 2] = o.z;
        }
        carry += tile_total;
        if (last_tile) {
            if (threadIdx.x == 0) {
                info->start[c] = class_start;
                info->count[c] = carry - class_start;
            }
            class_start = carry;
        }
    }
    if (threadIdx.x == 0) {
        const bool sane = carry <= previous->total;
        // classes without tiles: empty segments where they belong in the order (ascending starts)
        uint32_t start = 0;
        for (int c = 0; c < N_CLASSES; c++) {
            if ((present >> c) & 1u) {
                if (!sane) info->count[c] = 0;
                start = info->start[c] + info->count[c];
            } else {
                info->start[c] = start;
                info->count[c] = 0;
            }
        }
        info->total = sane ? carry : 0;
        info->cursor = 0;
    }
}

// Writes the class-sorted index list: a record that continues with class c goes to
// block_offsets[c][chunk] + (number of class-c records before it in its chunk).  Records of paths
// that ended are left out.  One block per chunk of the pool `previous` describes.  The block is the only reader of its
// chunk's offsets and leaves zeros there: the next bounce adds its class counts to them (add_chunk_counts).
__global__ __launch_bounds__(BLOCK) void k_bin_scatter(PathPool pool, const BinInfo* previous, uint32_t* block_offsets,
                                                       uint32_t count_stride, uint32_t class_mask, uint32_t rounds, uint32_t* list_out) {
    const ChunkTable table = chunk_table(previous, rounds * BLOCK);
    if (blockIdx.x >= table.n_chunks) return;
    const uint32_t capacity = previous->total; // the list is no longer than the pool it sorts; nothing is written past that
    int chunk_class;
    uint32_t first, n;
    locate_chunk(table, rounds * BLOCK, blockIdx.x, chunk_class, first, n);
    __shared__ uint32_t counts[BIN_ROUNDS][BLOCK / 64][N_CLASSES];
    const int wave = threadIdx.x >> 6;
    // the chunk's offsets are asked for first, so that they arrive while the classes are read and counted
    const bool owns_class = threadIdx.x < N_CLASSES && ((class_mask >> threadIdx.x) & 1u);
    uint32_t chunk_offset = 0;
    if (owns_class) {
        chunk_offset = block_offsets[threadIdx.x * count_stride + blockIdx.x];
        block_offsets[threadIdx.x * count_stride + blockIdx.x] = 0;
    }
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
    if (owns_class) {
        uint32_t running = chunk_offset;
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
        if (classes[r] >= 0) {
            const uint32_t place = counts[r][wave][classes[r]] + below[r];
            if (place < capacity) list_out[place] = first + r * BLOCK + threadIdx.x;
        }
}

// Completes the records of the camera rays that waited for k_traverse: where the ray ends (the closest hit of the
// object scan, overridden where a triangle is closer, or the free-flight distance k_generate left in the record),
// and the class of that vertex; a ray that leaves the scene takes the background colour.
__global__ __launch_bounds__(BLOCK) void k_primary_resume(DevScene sc, PathPool pool, SampleResults res, DeferQueues queues) {
    const uint32_t n = queues.counters[1];
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const uint32_t sid = queues.vertices[i];
        const RayResult found = queues.results[2 * sid + 1];
        const uint4 hit = ldu4(pool, PL_HIT, sid), ids = ldu4(pool, PL_IDS, sid);
        const double scatter_distance = __hiloint2double(static_cast<int>(hit.y), static_cast<int>(hit.x));
        HitInfo h;
        no_hit(h);
        if (found.obj >= 0) {
            const DevObject& o = sc.objects[found.obj];
            h.t = found.t;
            h.obj = found.obj;
            h.prim = o.type == PTK_OBJ_MESH ? found.prim : (o.type == PTK_OBJ_TRIANGLE ? static_cast<int>(o.first_tri) : 0);
            h.cls = o.mat.type;
        }
        const Vertex v = vertex_of(h, scatter_distance);
        if (v.cls < 0) {
            flush_color(res, sid, sc.background[0], sc.background[1], sc.background[2], false);
            pool.cls[sid] = CLASS_DONE;
        }
        else {
            stu4(pool, PL_HIT, sid, static_cast<uint32_t>(__double2loint(v.t)), static_cast<uint32_t>(__double2hiint(v.t)), hit.z, hit.w);
            stu4(pool, PL_IDS, sid, ids.x, static_cast<uint32_t>(v.obj), static_cast<uint32_t>(v.prim), 0u);
            pool.cls[sid] = static_cast<uint8_t>(v.cls);
        }
    }
}

// Counts, for every chunk of the list `info` describes, the classes of the records the bounce wrote for it:
// block_counts[class][chunk], the input of the next class sort.  Scenes with meshes use it: there two kernels write
// the records of a chunk (k_bounce, and k_bounce_resume for the vertices that waited for k_traverse), and counting
// in k_bounce_resume took one atomic per wave, class and chunk on counters that share a few cache lines at any
// moment, which the atomic unit of one L2 channel serialises.  One block per chunk; a pool's class bytes are 1 byte per record.
__global__ __launch_bounds__(BLOCK) void k_count_classes(PathPool out, const BinInfo* info, uint32_t class_mask, uint32_t rounds,
                                                         uint32_t* block_counts, uint32_t count_stride) {
    __shared__ uint32_t wave_counts[BLOCK / 64][N_CLASSES];
    const ChunkTable table = chunk_table(info, rounds * BLOCK);
    for (uint32_t chunk = blockIdx.x; chunk < table.n_chunks; chunk += gridDim.x) {
        int chunk_class;
        uint32_t first, n;
        locate_chunk(table, rounds * BLOCK, chunk, chunk_class, first, n);
        uint32_t my_counts[N_CLASSES];
#pragma unroll
        for (int c = 0; c < N_CLASSES; c++) my_counts[c] = 0;
        for (uint32_t qi = threadIdx.x; qi - threadIdx.x < n; qi += BLOCK) { // uniform trip count: the ballots need every lane
            const int cls = qi < n ? static_cast<int>(out.cls[first + qi]) : -1;
#pragma unroll
            for (int c = 0; c < N_CLASSES; c++)
                if ((class_mask >> c) & 1u) my_counts[c] += static_cast<uint32_t>(__popcll(__ballot(cls == c)));
        }
        store_chunk_counts(wave_counts, my_counts, block_counts, count_stride, chunk);
    }
}

// Closest triangle for every queued ray: the meshes of the scene in object order, each hierarchy walked with
// the shrinking t_max.  One ray per lane, and a wave that has lost enough of its rays takes new ones from the
// queue: the number of nodes a ray visits varies widely (a few for most, hundreds for rays that graze a mesh),
// and a wave that kept its 64 rays until the last one finished ran mostly empty.  The grid is sized to fill the
// device once; every wave leaves when the queue is empty and its own rays are done.
//
// Every pass of the loop takes each ray one step further (walk_step, geometry.h): the box tests of an inner node's
// four children, or one triangle of a leaf, and 50 of 64 lanes do work in it.  (Measured on one MI355X, 1,048,576-triangle workload, per 8-spp frame: 22.6 ms, against 31 ms for a
// descent through inner nodes until every lane holds a leaf followed by a loop over the leaves, where 12 of 64
// lanes are active per box test, and 24.6 ms for one step of the more numerous kind per pass.)
// queues.counters: [0] rays queued, [2] next ray to hand out (zero at launch).
#ifndef PTK_REFILL_LANES
#define PTK_REFILL_LANES 16
#endif
constexpr int REFILL_LANES = PTK_REFILL_LANES; // idle lanes at which a wave fetches

__global__ __launch_bounds__(BLOCK) PTK_WAVES_PER_SIMD(PTK_TRAVERSE_WAVES) void k_traverse(DevScene sc, DeferQueues queues,
                                                                                           unsigned long long* statistics) {
    __shared__ StackEntry lds_stack[LDS_STACK_DEPTH * BLOCK];
    __shared__ uint32_t wave_sums[BLOCK / 64][N_STATISTICS];
    PTK_TRAVERSAL_STACK(st, lds_stack, sc);
    const uint32_t n = queues.counters[0];
    // the list of meshes, in LDS when it fits: a ray consults it whenever it moves on to its next mesh
    __shared__ DevMeshRef lds_meshes[TRAVERSE_LDS_MESHES];
    const bool meshes_in_lds = sc.n_meshes <= TRAVERSE_LDS_MESHES;
    if (meshes_in_lds) {
        for (int i = threadIdx.x; i < sc.n_meshes; i += BLOCK) lds_meshes[i] = sc.meshes[i];
        __syncthreads();
    }
    uint32_t stat[N_STATISTICS] = {}; // [0] rays, [1] nodes visited, [2] triangles tested
    bool exhausted = false;           // wave-uniform: the queue has been handed out completely
    // A wave reserves a chunk of the queue at a time and hands its rays to lanes as they fall idle: the queue's counter
    // is one address, which serves about a hundred atomics per microsecond for the whole device (one atomic per fetch
    // of 16 to 64 rays, three million per frame, took as long as the walks themselves).  The chunk shrinks with the
    // queue so that a short queue is still spread over all waves.
    const uint32_t chunk = min(max(n / (gridDim.x * (BLOCK / 64) * 8u), 32u), 512u);
    uint32_t reserved_next = 0, reserved_end = 0; // wave-uniform: rays reserved and not yet handed out
    bool has_ray = false;
    V3 org = v3(0.0);
    RayPrep rp{};
    SlabRay slab{};
    RayResult best{INF, -1, 0};
    uint32_t slot = 0;
    int mesh = 0, object = 0;   // the mesh being walked: position in the list of meshes, object index
    int64_t first_tri = 0;
    bool any_hit = false;
    int32_t node = BVH_EMPTY;   // BVH_EMPTY: the walk of the current mesh is over (or has not begun), or the lane has no ray
    int tri_i = 0;              // the triangle of the lane's leaf that is tested next
#ifdef PTK_TRAVERSE_PROBE
    uint32_t ray_steps = 0;
#endif
    // A lane whose walk of a mesh is over (or has not begun) moves on to the ray's next mesh, or hands in the result.
    auto next_mesh = [&]() {
        if (has_ray && node == BVH_EMPTY) {
            mesh++;
            if (mesh < sc.n_meshes) {
                const DevMeshRef m = meshes_in_lds ? lds_meshes[mesh] : sc.meshes[mesh];
                object = m.object;
                node = m.bvh_root;
                first_tri = m.first_tri;
                // A shadow ray (even slot) is blocked by any triangle of a mesh that lets no connection
                // through: the march ends in black whichever triangle it meets first.
                any_hit = (slot & 1u) == 0 && m.opaque != 0;
                st.sp = 0;
            }
            else {
                queues.results[slot] = best;
                has_ray = false;
#ifdef PTK_TRAVERSE_PROBE
                stat[4] += ray_steps > 128 ? 1 : 0; // rays by the number of passes they took
                stat[5] += ray_steps > 512 ? 1 : 0;
                stat[6] += ray_steps > 2048 ? 1 : 0;
                ray_steps = 0;
#endif
            }
        }
    };
    for (;;) {
        next_mesh();
        // new rays for the idle lanes, once there are enough of them
        const unsigned long long idle = __ballot(!has_ray);
        const uint32_t n_idle = static_cast<uint32_t>(__popcll(idle));
        if (!exhausted && n_idle >= REFILL_LANES) {
            if (reserved_next >= reserved_end) { // reserve the next chunk of the queue
                uint32_t base = 0;
                if ((threadIdx.x & 63) == 0) base = atomicAdd(queues.counters + 2, chunk);
                base = __builtin_amdgcn_readfirstlane(base);
                reserved_next = min(base, n);
                reserved_end = min(base + chunk, n);
                exhausted = base >= n;
            }
            if (!has_ray) {
                const uint32_t i = reserved_next + lanes_below(idle);
                if (i < reserved_end) {
                    const RayRequest q = queues.rays[i];
                    org = v3(q.ox, q.oy, q.oz);
                    const V3 dir = v3(q.dx, q.dy, q.dz);
                    rp = prepare_ray(dir);
                    slab = slab_ray(org, dir, sc.mesh_bound_abs);
                    best = RayResult{q.limit, static_cast<int32_t>(q.object), 0}; // the closest hit among the other objects
                    slot = q.slot;
                    mesh = -1;
                    node = BVH_EMPTY;
                    has_ray = true;
                    stat[0]++;
                }
            }
            reserved_next = min(reserved_next + n_idle, reserved_end);
            next_mesh(); // the new rays enter their first mesh
        }
        if (__ballot(has_ray) == 0) {
            if (exhausted) break;
            continue; // all 64 lanes are idle: the next pass fetches
        }
#ifdef PTK_TRAVERSE_PROBE
        if (lanes_below(__ballot(true)) == 0) stat[3]++; // passes of the wave
        stat[7] += has_ray ? 1 : 0;                      // lanes that held a ray in them
#endif
        if (node != BVH_EMPTY) { // (a lane without a ray holds BVH_EMPTY too)
#ifdef PTK_TRAVERSE_PROBE
            ray_steps++;
#endif
            const WalkStep step = walk_step<true>(sc, node, tri_i, first_tri, org, rp, slab, best.t, st);
            if (step.found) {
                best.obj = object;
                best.prim = step.prim;
                if (any_hit) { // nothing else matters for this ray
                    mesh = sc.n_meshes;
                    node = BVH_EMPTY;
                    tri_i = 0;
                }
            }
        }
    }
    stat[1] = st.nodes_visited;
    stat[2] = st.triangles_tested;
    add_block_statistics(statistics, wave_sums, stat);
}

// The same with four lanes per ray (geometry.h, "four lanes per ray"): a wave holds 16 rays; at an inner node each of a
// ray's lanes tests one child box, at a leaf one triangle.  Per node a ray then costs two 16-byte loads per lane that the
// four lanes aim at one 128-byte line, instead of seven loads of one lane.  A quad whose ray is finished takes a new one
// from the queue once REFILL_QUADS of the wave's 16 are idle.
// Measured on one MI355X, 1,048,576-triangle workload, per two steps: 84 ms against the 65 ms of k_traverse (so that one
// stays the default; ptk_set_option "traverse_lanes" selects).  The rays of a wave are in different phases of their
// walks here as there (15.4 of 64 lanes, i.e. four of sixteen quads, are active per vector instruction, 11.6 lanes in
// k_traverse), and a wave of 16 rays has a quarter of the loads in flight; 4 waves per SIMD: 96 ms, 6: 85 ms, 7 and 8
// (with registers spilled): 112 and 153 ms; refill at 2, 4 or 8 idle quads: no difference.
#ifndef PTK_REFILL_QUADS
#define PTK_REFILL_QUADS 4
#endif
#ifndef PTK_QUAD_RAY_CHUNK
#define PTK_QUAD_RAY_CHUNK 32
#endif
constexpr int REFILL_QUADS = PTK_REFILL_QUADS;
// A wave reserves this many rays of the queue at a time and hands them to its quads as they fall idle: one atomic on
// the queue's counter per chunk (that one address serves about a hundred atomics per microsecond, and a wave of 16
// rays would otherwise ask several times as often as one of 64).
constexpr uint32_t QUAD_RAY_CHUNK = PTK_QUAD_RAY_CHUNK;
constexpr int QUAD_RAYS_PER_BLOCK = BLOCK / 4;

__global__ __launch_bounds__(BLOCK) PTK_WAVES_PER_SIMD(PTK_TRAVERSE_QUAD_WAVES) void k_traverse_quad(DevScene sc, DeferQueues queues,
                                                                                                     unsigned long long* statistics) {
    __shared__ StackEntry lds_stack[QUAD_STACK_DEPTH * QUAD_RAYS_PER_BLOCK];
    __shared__ uint32_t wave_sums[BLOCK / 64][N_STATISTICS];
    __shared__ DevMeshRef lds_meshes[TRAVERSE_QUAD_LDS_MESHES];
    const uint32_t n = queues.counters[0];
    const bool meshes_in_lds = sc.n_meshes <= TRAVERSE_QUAD_LDS_MESHES;
    if (meshes_in_lds) {
        for (int i = threadIdx.x; i < sc.n_meshes; i += BLOCK) lds_meshes[i] = sc.meshes[i];
        __syncthreads();
    }
    const int c = threadIdx.x & 3; // this lane's child / triangle
    QuadStack st;
    st.column = (LdsStackEntry*) lds_stack + (threadIdx.x >> 2);
    st.stride = QUAD_RAYS_PER_BLOCK;
    st.sp = 0;
    st.overflows = sc.stack_overflows;
    uint32_t stat[N_STATISTICS] = {}; // [0] rays, [1] nodes visited, [2] triangles tested (counted by lane 0 of each quad)
    bool exhausted = false;           // wave-uniform: the queue has been handed out completely
    uint32_t reserved_next = 0, reserved_end = 0; // wave-uniform: the rays this wave has reserved and not yet handed out
    bool has_ray = false;             // everything from here on is the same in the four lanes of a quad
    V3 org = v3(0.0);
    RayPrep rp{};
    SlabRay1 slab{};
    RayResult best{INF, -1, 0};
    uint32_t slot = 0;
    int mesh = 0, object = 0;
    int64_t first_tri = 0;
    bool any_hit = false;
    int32_t node = BVH_EMPTY;
    for (;;) {
        const unsigned long long idle = __ballot(!has_ray);
        const uint32_t n_idle = static_cast<uint32_t>(__popcll(idle)) / 4u; // quads
        if (!exhausted && n_idle >= REFILL_QUADS) {
            if (reserved_next >= reserved_end) { // reserve the next chunk of the queue
                uint32_t base = 0;
                if ((threadIdx.x & 63) == 0) base = atomicAdd(queues.counters + 2, QUAD_RAY_CHUNK);
                base = __builtin_amdgcn_readfirstlane(base);
                reserved_next = min(base, n);
                reserved_end = min(base + QUAD_RAY_CHUNK, n);
                exhausted = base >= n;
            }
            if (!has_ray) {
                const uint32_t i = reserved_next + lanes_below(idle) / 4u;
                if (i < reserved_end) {
                    const RayRequest q = queues.rays[i]; // the four lanes read the same 64 bytes
                    org = v3(q.ox, q.oy, q.oz);
                    const V3 dir = v3(q.dx, q.dy, q.dz);
                    rp = prepare_ray(dir);
                    slab = slab_ray1(org, dir, sc.mesh_bound_abs);
                    best = RayResult{q.limit, static_cast<int32_t>(q.object), 0};
                    slot = q.slot;
                    mesh = -1;
                    node = BVH_EMPTY;
                    has_ray = true;
                    if (c == 0) stat[0]++;
                }
            }
            reserved_next = min(reserved_next + n_idle, reserved_end);
        }
        if (__ballot(has_ray) == 0) {
            if (exhausted) break;
            continue;
        }
        if (has_ray && node == BVH_EMPTY) { // between meshes: on to the ray's next mesh, or the ray is finished
            mesh++;
            if (mesh < sc.n_meshes) {
                const DevMeshRef m = meshes_in_lds ? lds_meshes[mesh] : sc.meshes[mesh];
                object = m.object;
                node = m.bvh_root;
                first_tri = m.first_tri;
                any_hit = (slot & 1u) == 0 && m.opaque != 0; // see k_traverse
                st.sp = 0;
            }
            else {
                if (c == 0) queues.results[slot] = best;
                has_ray = false;
            }
        }
        while (node >= 0) {
            if (c == 0) stat[1]++;
            node = visit_inner_quad(sc.nodes_quad, node, c, slab, best.t, st);
        }
        while (node < 0 && node != BVH_EMPTY) {
            if (c == 0) stat[2] += (static_cast<uint32_t>(~node) & 7u) + 1u;
            int prim = -1;
            const bool found = visit_leaf_quad(sc, node, first_tri, c, org, rp, best.t, prim);
            node = pop_closer_quad(st, best.t);
            if (found) {
                best.obj = object;
                best.prim = prim;
                if (any_hit) {
                    mesh = sc.n_meshes;
                    node = BVH_EMPTY;
                }
            }
        }
    }
    add_block_statistics(statistics, wave_sums, stat);
}

// Adds the per-block statistics slots to totals[N_STATISTICS] (zero before): one slot (64 bytes) per thread, a
// reduction inside each wave, one atomic per wave and value.
__global__ __launch_bounds__(BLOCK) void k_sum_statistics(const unsigned long long* slots, unsigned long long* totals) {
    static_assert(N_STATISTICS == 8 && STATISTICS_SLOTS % BLOCK == 0, "a slot is read as four pairs, one slot per thread");
    const ulonglong2* slot = reinterpret_cast<const ulonglong2*>(slots + static_cast<size_t>(blockIdx.x * BLOCK + threadIdx.x) * N_STATISTICS);
    unsigned long long sum[N_STATISTICS];
#pragma unroll
    for (int k = 0; k < N_STATISTICS / 2; k++) {
        const ulonglong2 v = slot[k];
        sum[2 * k] = v.x;
        sum[2 * k + 1] = v.y;
    }
#pragma unroll
    for (int k = 0; k < N_STATISTICS; k++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum[k] += __shfl_down(sum[k], off);
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < N_STATISTICS; k++)
            if (sum[k]) atomicAdd(totals + k, sum[k]);
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
            double2 s0, s1, s2;
            load_surface(res, sid, cb, s0, s1, s2);
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
        const size_t idx = batch_pixel(b, p);
        const double n_prev = static_cast<double>(pixel_counters[idx]);
        for (int j = 0; j < 3; j++) {
            image[3 * idx + j] = (image[3 * idx + j] * n_prev + (sum[j] / n_add) * n_add) / (n_prev + n_add);
            position[3 * idx + j] = (position[3 * idx + j] * n_prev + (sum[3 + j] / n_add) * n_add) / (n_prev + n_add);
            normal[3 * idx + j] = (normal[3 * idx + j] * n_prev + (sum[6 + j] / n_add) * n_add) / (n_prev + n_add);
        }
        pixel_counters[idx] += b.n_samples;
    }
}

// Folds the film `from` into the film `into`, pixel by pixel, as raytrace_section folds a job into its buffers
// (increment_average, app/pathtracer.cpp:232-235): the sample-count-weighted mean of the two running means.  A pixel
// that only one of the films has samples for keeps that film's bits, so films of disjoint pixels merge exactly.
__global__ __launch_bounds__(BLOCK) void k_merge_films(size_t n_pixels, double* image, double* position, double* normal,
                                                       unsigned long long* counters, const double* from_image,
                                                       const double* from_position, const double* from_normal,
                                                       const unsigned long long* from_counters) {
    for (size_t p = static_cast<size_t>(blockIdx.x) * BLOCK + threadIdx.x; p < n_pixels; p += static_cast<size_t>(gridDim.x) * BLOCK) {
        const unsigned long long add = from_counters[p];
        if (add == 0) continue;
        const unsigned long long prev = counters[p];
        const double n_prev = static_cast<double>(prev), n_add = static_cast<double>(add);
        for (size_t c = 3 * p; c < 3 * p + 3; c++) {
            image[c] = prev == 0 ? from_image[c] : (image[c] * n_prev + from_image[c] * n_add) / (n_prev + n_add);
            position[c] = prev == 0 ? from_position[c] : (position[c] * n_prev + from_position[c] * n_add) / (n_prev + n_add);
            normal[c] = prev == 0 ? from_normal[c] : (normal[c] * n_prev + from_normal[c] * n_add) / (n_prev + n_add);
        }
        counters[p] = prev + add;
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
        double2 s0, s1, s2;
        load_surface(res, sid, reinterpret_cast<const double2*>(res.color)[2 * static_cast<size_t>(sid) + 1], s0, s1, s2);
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
    PTK_TRAVERSAL_STACK(st, lds_stack, sc);
    for (uint32_t p = blockIdx.x * BLOCK + threadIdx.x; p < b.n_pixels; p += gridDim.x * BLOCK) {
        const uint32_t pixel = batch_pixel(b, p);
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

// Self-test of the record prefetch through LDS (request_record / staged_record, the pair k_bounce stages its
// records with): record list[i] of `in` travels memory -> LDS -> registers -> record i of `out`, in the same
// two-step pipeline as k_bounce (the request for the next round is issued in the middle of the current one), with
// some lanes switched off and unrelated vector loads in flight, so a missing or wrong wait count between the
// asynchronous load into LDS and the LDS read shows up as a record that differs from its source.
__global__ __launch_bounds__(BLOCK) void k_selftest_staging(PathPool in, PathPool out, const uint32_t* list, uint32_t n,
                                                            const double* noise, uint32_t noise_mask, double* sink) {
    __shared__ double2 lds_record[N_PLANES * BLOCK];
    double2* const wave_stage = lds_record + 64 * __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double acc = 0;
    const uint32_t stride = gridDim.x * BLOCK;
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    auto active = [&](uint32_t k) { return k < n && (k % 5u) != 3u; };
    if (active(i)) request_record(in, list[i], wave_stage);
    wait_for_loads();
    for (; i - threadIdx.x < n; i += stride) { // block-uniform trip count
        Record cur{};
        if (active(i)) cur = staged_record(lds_record);
        acc += noise[(i * 2654435761u) & noise_mask]; // a load that is still in flight when the next request is issued
        if (active(i + stride)) request_record(in, list[i + stride], wave_stage);
        acc += noise[(i * 40503u + 17u) & noise_mask];
        wait_for_loads(); // before the stores, as in k_bounce; the top of the next round reads the staged record unguarded
        if (active(i)) {
            st2(out, PL_RAY0, i, cur.r0.x, cur.r0.y);
            st2(out, PL_RAY1, i, cur.r1.x, cur.r1.y);
            st2(out, PL_RAY2, i, cur.r2.x, cur.r2.y);
            st2(out, PL_THR0, i, cur.t0.x, cur.t0.y);
            st2(out, PL_THR1, i, cur.t1.x, cur.t1.y);
            st2(out, PL_SAV0, i, cur.v0.x, cur.v0.y);
            st2(out, PL_SAV1, i, cur.v1.x, cur.v1.y);
            st2(out, PL_COL, i, cur.cc.x, cur.cc.y);
            stu4(out, PL_HIT, i, cur.hit.x, cur.hit.y, cur.hit.z, cur.hit.w);
            stu4(out, PL_IDS, i, cur.ids.x, cur.ids.y, cur.ids.z, cur.ids.w);
        }
    }
    sink[blockIdx.x * BLOCK + threadIdx.x] = acc;
}

void launch_selftest_staging(hipStream_t s, uint32_t blocks, const PathPool& in, const PathPool& out, const uint32_t* list, uint32_t n,
                             const double* noise, uint32_t noise_mask, double* sink) {
    hipLaunchKernelGGL(k_selftest_staging, dim3(blocks), dim3(BLOCK), 0, s, in, out, list, n, noise, noise_mask, sink);
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
                     BinInfo* first_info, uint32_t* block_counts, uint32_t count_stride, const DeferQueues& queues) {
    const dim3 grid(min(bin_blocks(b.n_paths), static_cast<uint32_t>(STATISTICS_SLOTS))), block(BLOCK);
    if (sc.has_meshes) hipLaunchKernelGGL((k_generate<true, true>), grid, block, 0, s, sc, b, pool, res, first_info, block_counts, count_stride, queues);
    else if (sc.has_triangles) hipLaunchKernelGGL((k_generate<true, false>), grid, block, 0, s, sc, b, pool, res, first_info, block_counts, count_stride, queues);
    else hipLaunchKernelGGL((k_generate<false, false>), grid, block, 0, s, sc, b, pool, res, first_info, block_counts, count_stride, queues);
}

void launch_bin(hipStream_t s, const DevScene& sc, const PathPool& pool, uint32_t bound, uint32_t rounds, const BinInfo* previous,
                uint32_t* block_counts, uint32_t count_stride, BinInfo* info, uint32_t* list_out) {
    hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(1024), 0, s, block_counts, count_stride, sc.class_mask, rounds, previous, info);
    hipLaunchKernelGGL(k_bin_scatter, dim3(chunk_blocks(bound, rounds)), dim3(BLOCK), 0, s, pool, previous, block_counts, count_stride,
                       sc.class_mask, rounds, list_out);
}

// ---- choice of kernel variant (see variants.h) ---------------------------------------------------
namespace {
struct Variant {
    bool tris, uv, defer, pass, cache;
};
Variant variant_of(const DevScene& sc, bool specialised) {
    Variant v{sc.has_triangles != 0, sc.needs_uv != 0, sc.has_meshes != 0, sc.has_pass_through != 0, specialised && sc.tables_fit_lds != 0};
    if (!v.cache) v.tris = v.uv = v.pass = true;  // the general kernels
    else if (v.tris) v.pass = v.pass || !v.defer; // scenes with triangles: PASS = false exists for the deferring kernels only
    return v;
}
const char* tf(bool v) {
    return v ? "true" : "false";
}
} // namespace

std::string bounce_kernel_name(const DevScene& sc, bool specialised) {
    const Variant v = variant_of(sc, specialised);
    return std::string("ptk::k_bounce<") + tf(v.tris) + ", " + tf(v.uv) + ", " + tf(v.defer) + ", " + tf(v.pass) + ", " + tf(v.cache) + ">";
}

void launch_bounce(const LaunchContext& lc, uint32_t bound, uint32_t rounds, const DevScene& sc, const BatchParams& b, const PathPool& in,
                   const PathPool& out, const SampleResults& res, const uint32_t* list, const BinInfo* info, uint32_t* block_counts,
                   uint32_t count_stride, unsigned long long* statistics, const DeferQueues& queues) {
    const hipStream_t s = lc.stream;
    const dim3 grid(min(chunk_blocks(bound, rounds), static_cast<uint32_t>(STATISTICS_SLOTS)));
    const Variant v = variant_of(sc, lc.specialised);
    // the kernels that finish every vertex themselves add their class counts to the zeros k_bin_scatter left (add_chunk_counts)
#define PTK_CASE(T, U, D, P, C)                                                                                                    \
    if (v.tris == T && v.uv == U && v.defer == D && v.pass == P && v.cache == C)                                                   \
        return launch_bounce_variant<T, U, D, P, C>(s, grid, sc, b, in, out, res, list, info, rounds, block_counts, count_stride, \
                                                    statistics, queues)
    PTK_CASE(false, false, false, false, true);
    PTK_CASE(false, false, false, true, true);
    PTK_CASE(false, true, false, false, true);
    PTK_CASE(false, true, false, true, true);
    PTK_CASE(true, false, false, true, true);
    PTK_CASE(true, true, false, true, true);
    PTK_CASE(true, false, true, false, true);
    PTK_CASE(true, false, true, true, true);
    PTK_CASE(true, true, true, false, true);
    PTK_CASE(true, true, true, true, true);
    PTK_CASE(true, true, false, true, false);
    PTK_CASE(true, true, true, true, false);
#undef PTK_CASE
}

void launch_tail(const LaunchContext& lc, uint32_t bound, const DevScene& sc, const BatchParams& b, const PathPool& in, const PathPool& out,
                 const SampleResults& res, const uint32_t* list, BinInfo* info, unsigned long long* statistics) {
    const hipStream_t s = lc.stream;
    // at most one device fill (two blocks of four waves per CU at two waves per SIMD): the waves are persistent
    const dim3 grid(min((bound + BLOCK - 1) / BLOCK + 1, 2u * lc.compute_units));
    Variant v = variant_of(sc, lc.specialised);
    if (v.tris) v.pass = true;
#define PTK_CASE(T, U, P, C)                                            \
    if (v.tris == T && v.uv == U && v.pass == P && v.cache == C)        \
        return launch_tail_variant<T, U, P, C>(s, grid, sc, b, in, out, res, list, info, statistics)
    PTK_CASE(false, false, false, true);
    PTK_CASE(false, false, true, true);
    PTK_CASE(false, true, false, true);
    PTK_CASE(false, true, true, true);
    PTK_CASE(true, false, true, true);
    PTK_CASE(true, true, true, true);
    PTK_CASE(true, true, true, false);
#undef PTK_CASE
}

void launch_traverse(const LaunchContext& lc, uint32_t bound, const DevScene& sc, const DeferQueues& queues, unsigned long long* statistics) {
    const hipStream_t s = lc.stream;
    if (lc.quad_traverse) {
        // one device fill of waves that hold 16 rays each
        const uint32_t resident = lc.compute_units * PTK_TRAVERSE_QUAD_WAVES;
        const dim3 grid(min((2 * bound + QUAD_RAYS_PER_BLOCK - 1) / QUAD_RAYS_PER_BLOCK + 1, resident));
        hipLaunchKernelGGL(k_traverse_quad, grid, dim3(BLOCK), 0, s, sc, queues, statistics);
        return;
    }
    // enough blocks to fill the device once (PTK_TRAVERSE_WAVES waves per SIMD, four SIMDs per CU, four waves per block)
    const uint32_t resident = lc.compute_units * PTK_TRAVERSE_WAVES;
    const dim3 grid(min((2 * bound + BLOCK - 1) / BLOCK + 1, resident));
    hipLaunchKernelGGL(k_traverse, grid, dim3(BLOCK), 0, s, sc, queues, statistics);
}

void launch_bounce_resume(const LaunchContext& lc, uint32_t bound, const DevScene& sc, const BatchParams& b, const PathPool& out,
                          const SampleResults& res, unsigned long long* statistics, const DeferQueues& queues) {
    const hipStream_t s = lc.stream;
    const dim3 grid(min((bound + BLOCK - 1) / BLOCK + 1, static_cast<uint32_t>(STATISTICS_SLOTS)));
    const Variant v = variant_of(sc, lc.specialised);
#define PTK_CASE(U, P, C)                                 \
    if (v.uv == U && v.pass == P && v.cache == C)         \
        return launch_resume_variant<U, P, C>(s, grid, sc, b, out, res, statistics, queues)
    PTK_CASE(false, false, true);
    PTK_CASE(false, true, true);
    PTK_CASE(true, false, true);
    PTK_CASE(true, true, true);
    PTK_CASE(true, true, false);
#undef PTK_CASE
}

void launch_primary_resume(hipStream_t s, uint32_t n_paths, const DevScene& sc, const PathPool& pool, const SampleResults& res,
                           const DeferQueues& queues) {
    hipLaunchKernelGGL(k_primary_resume, dim3(min((n_paths + BLOCK - 1) / BLOCK, 8192u)), dim3(BLOCK), 0, s, sc, pool, res, queues);
}

void launch_count_classes(const LaunchContext& lc, uint32_t bound, uint32_t rounds, const DevScene& sc, const PathPool& out, const BinInfo* info,
                          uint32_t* block_counts, uint32_t count_stride) {
    hipLaunchKernelGGL(k_count_classes, dim3(chunk_blocks(bound, rounds)), dim3(BLOCK), 0, lc.stream, out, info, sc.class_mask, rounds,
                       block_counts, count_stride);
}

void launch_sum_statistics(hipStream_t s, const unsigned long long* slots, unsigned long long* totals) {
    hipLaunchKernelGGL(k_sum_statistics, dim3(STATISTICS_SLOTS / BLOCK), dim3(BLOCK), 0, s, slots, totals);
}

void launch_accumulate(hipStream_t s, const BatchParams& b, const SampleResults& res, double* image, double* position, double* normal,
                       unsigned long long* pixel_counters) {
    hipLaunchKernelGGL(k_accumulate, dim3(grid_for(b.n_pixels, 4096)), dim3(BLOCK), 0, s, b, res, image, position, normal,
                       pixel_counters);
}

void launch_merge_films(hipStream_t s, size_t n_pixels, double* image, double* position, double* normal, unsigned long long* counters,
                        const double* from_image, const double* from_position, const double* from_normal,
                        const unsigned long long* from_counters) {
    const uint32_t blocks = static_cast<uint32_t>(std::min<size_t>((n_pixels + BLOCK - 1) / BLOCK, 8192));
    hipLaunchKernelGGL(k_merge_films, dim3(std::max(blocks, 1u)), dim3(BLOCK), 0, s, n_pixels, image, position, normal, counters,
                       from_image, from_position, from_normal, from_counters);
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
