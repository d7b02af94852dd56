// Wavefront path-tracing kernels for gfx950.
//
//   k_generate     camera samples -> path records              (compute_pixel_color + Camera::make_ray)
//   k_trace        per record: the visibility march of the connection the last shade left pending
//                  (compute_visibility), then the free-flight distance and the closest hit of the
//                  new ray (raytrace's loop head + find_closest_hit); records the class of the vertex
//                  and leaves per-block class counts; paths that ended are flushed to the results
//   k_bin_scan     exclusive scan of the per-block class counts
//   k_bin_scatter  writes the class-sorted index list           (stable: record order is preserved)
//   k_shade<C>     one launch per material class over its segment of the list: emission with MIS,
//                  next-event sampling up to the connection record, BRDF sampling, medium switch,
//                  Russian roulette (raytrace's loop body); reads the record through the list and
//                  writes the updated record to the other pool at the list position
//   k_accumulate   per-pixel sample means folded into the film  (compute_pixel_color's division and
//                  increment_average)
//
// One lane owns one path.  Records move once per bounce, from one pool to the other, which keeps
// the records in flight packed (coalesced 16-byte accesses whatever fraction of the batch is still
// alive) at no extra traffic: a shade kernel rewrites most of the record anyway.  Sorting by class
// uses block-local ballots and a scan instead of global atomics: with one atomic per wave on a
// shared counter the kernels ran at the rate of that one cache line (about 100 atomics per
// microsecond) instead of at the rate of the arithmetic.  All arithmetic is fp64, as in the
// reference; MFMA has no role here (divergent scalar-per-lane work, no dense contraction).
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "shading.h"

namespace ptk {

constexpr int BLOCK = 256;
// Register budgets: the kernels are compiled for a fixed number of waves per SIMD (512 registers
// per lane and SIMD are shared by the resident waves).
#ifndef PTK_TRACE_WAVES
#define PTK_TRACE_WAVES 4
#endif
#ifndef PTK_SHADE_WAVES
#define PTK_SHADE_WAVES 3
#endif
#define PTK_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n, n)))
static_assert(BLOCK == BIN_BLOCK, "k_extend and the bin kernels share their block geometry");

PTK_DEV TraversalStack make_stack(StackEntry* lds) {
    TraversalStack st;
    st.lds = lds + threadIdx.x;
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
PTK_DEV double key_as_double(uint64_t k) { return __longlong_as_double(static_cast<long long>(k)); }
PTK_DEV uint64_t double_as_key(double d) { return static_cast<uint64_t>(__double_as_longlong(d)); }

// The path ended: its radiance goes to the per-sample results.
PTK_DEV void flush_color(const SampleResults& res, uint32_t sid, double r, double g, double bl) {
    double2* c = reinterpret_cast<double2*>(res.color) + 2 * static_cast<size_t>(sid);
    c[0] = make_double2(r, g);
    c[1] = make_double2(bl, 0.0);
}
PTK_DEV double2* surface_plane(const SampleResults& res, int plane) {
    return reinterpret_cast<double2*>(res.surface + static_cast<uint64_t>(plane) * res.plane_doubles);
}

// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_generate(DevScene sc, BatchParams b, PathPool pool, SampleResults res, BinInfo* first_info) {
    if (blockIdx.x == 0 && threadIdx.x == 0) first_info->total = b.n_paths; // records in the first pool
    for (uint32_t sid = blockIdx.x * BLOCK + threadIdx.x; sid < b.n_paths; sid += gridDim.x * BLOCK) {
        uint32_t pixel, sample;
        sample_coordinates(b, sid, pixel, sample);
        Rng rng{rng_key(b.seed, pixel, sample), 0};
        V3 org, dir;
        camera_ray(sc, pixel, rng, org, dir);
        st2(pool, PL_RAY0, sid, org.x, org.y);
        st2(pool, PL_RAY1, sid, org.z, dir.x);
        st2(pool, PL_RAY2, sid, dir.y, dir.z);
        st2(pool, PL_THR0, sid, 1.0, 1.0);
        st2(pool, PL_THR1, sid, 1.0, 0.0);
        st2(pool, PL_SAV0, sid, 0.0, 0.0);
        st2(pool, PL_SAV1, sid, 0.0, 0.0);
        st2(pool, PL_COL0, sid, 0.0, 0.0);
        st2(pool, PL_COL1, sid, 0.0, 0.0);
        stu4(pool, PL_IDS, sid, 0u | (RAY_TRANSMITTED << TYPE_SHIFT) | (static_cast<uint32_t>(sc.background_medium) << 16), sid,
             rng.ctr, 0u);
        pool.cls[sid] = 0;
        surface_plane(res, 0)[sid] = make_double2(0.0, 0.0);
        surface_plane(res, 1)[sid] = make_double2(0.0, 0.0);
        surface_plane(res, 2)[sid] = make_double2(0.0, 0.0);
    }
}

// -------------------------------------------------------------------------------------------------
// For every record of the current pool: finish the connection the last shade left pending
// (compute_visibility, src/objects.cpp:470-513, and the acceptance test of sample_light :558), then
// run the loop head of raytrace (app/pathtracer.cpp:53-67) for the new ray: free-flight distance,
// closest hit, miss handling.  Block b owns records [b * BIN_CHUNK, (b + 1) * BIN_CHUNK) and leaves
// its per-class counts in block_counts[class * count_stride + b].
// counters: [0] visibility-march segments, [1] extension rays, [2] connections marched.
// TRIS: the scene has triangles or meshes; UV: some material reads a map larger than 1x1 (texture
// coordinates are needed).  Scenes without them get kernels without that code and its registers.
template<bool TRIS, bool UV>
__global__ __launch_bounds__(BLOCK) PTK_WAVES_PER_SIMD(TRIS ? 2 : PTK_TRACE_WAVES) void k_trace(DevScene sc, BatchParams b, PathPool pool, SampleResults res, const BinInfo* previous,
                                                 uint32_t* block_counts, uint32_t count_stride, unsigned long long* counters) {
    const uint32_t n = previous->total; // records the previous bounce's shade kernels wrote
    if (blockIdx.x * BIN_CHUNK >= n) return;
    __shared__ StackEntry lds_stack[TRIS ? LDS_STACK_DEPTH * BLOCK : 1];
    __shared__ uint32_t wave_counts[BLOCK / 64][N_CLASSES];
    __shared__ uint32_t wave_rays[BLOCK / 64][3];
    TraversalStack st = make_stack(lds_stack);
    uint32_t my_counts[N_CLASSES];
#pragma unroll
    for (int c = 0; c < N_CLASSES; c++) my_counts[c] = 0;
    uint32_t segments = 0, extensions = 0, connections = 0; // per lane
    for (int r = 0; r < BIN_ROUNDS; r++) {
        const uint32_t idx = blockIdx.x * BIN_CHUNK + r * BLOCK + threadIdx.x;
        int cls = -1;
        const uint8_t mark = idx < n ? pool.cls[idx] : CLASS_DONE;
        if (mark != CLASS_DONE) {
            const uint4 ids = ldu4(pool, PL_IDS, idx);
            const double2 r0 = ld2(pool, PL_RAY0, idx), r1 = ld2(pool, PL_RAY1, idx);
            const V3 org = v3(r0.x, r0.y, r1.x);
            const uint32_t sid = ids.y;
            double2 c0 = make_double2(0.0, 0.0), c1 = c0;
            bool color_loaded = false, color_changed = false;
            if (ids.w & SHADOW_VALID) {
                c0 = ld2(pool, PL_COL0, idx);
                c1 = ld2(pool, PL_COL1, idx);
                color_loaded = true;
                const double2 s0 = ld2(pool, PL_SH0, idx), s1 = ld2(pool, PL_SH1, idx), s2 = ld2(pool, PL_SH2, idx);
                const double2 v1 = ld2(pool, PL_SAV1, idx);
                Rng rng{double_as_key(c1.y), 0};
                connections++;
                V3 visible = march_to_light<TRIS, UV>(sc, org, v3(s0.x, s0.y, s1.x), static_cast<int>((ids.w >> 16) & 0x7FFFu),
                                                    static_cast<int>(ids.w & 0xFFFFu), s1.y, rng, st, segments);
                if (visible.x != 0 || visible.y != 0 || visible.z != 0) {
                    c0.x += s2.x * visible.x;
                    c0.y += s2.y * visible.y;
                    c1.x += v1.y * visible.z;
                    color_changed = true;
                }
            }
            if (mark == CLASS_DEAD) {
                if (!color_loaded) {
                    c0 = ld2(pool, PL_COL0, idx);
                    c1 = ld2(pool, PL_COL1, idx);
                }
                flush_color(res, sid, c0.x, c0.y, c1.x);
            }
            else {
                const double2 r2 = ld2(pool, PL_RAY2, idx);
                const V3 dir = v3(r1.y, r2.x, r2.y);
                uint32_t medium = ids.x >> 16;
                if (medium == MEDIUM_NONE) medium = static_cast<uint32_t>(sc.background_medium);
                const DevMedium& med = sc.media[medium];
                double scatter_distance = INF;
                if (med.type == PTK_MEDIUM_HOMOGENEOUS_SCATTERING) { // HomogenousScatteringMedium::sample_distance
                    uint32_t pixel, sample;
                    sample_coordinates(b, sid, pixel, sample);
                    Rng rng{rng_key(b.seed, pixel, sample), ids.z};
                    int channel = rng.integer(0, 3);
                    double ext = med.extinction[channel];
                    if (ext != 0) scatter_distance = -log(rng.u01()) / ext;
                }
                HitInfo h;
                bool found = closest_hit<TRIS>(sc, org, dir, scatter_distance, h, st);
                extensions++;
                if (!found && scatter_distance == INF) {
                    // color += throughput * background_color; the path ends
                    if (!color_loaded) {
                        c0 = ld2(pool, PL_COL0, idx);
                        c1 = ld2(pool, PL_COL1, idx);
                    }
                    if (!sc.background_is_black) {
                        const double2 t0 = ld2(pool, PL_THR0, idx), t1 = ld2(pool, PL_THR1, idx);
                        c0.x += t0.x * sc.background[0];
                        c0.y += t0.y * sc.background[1];
                        c1.x += t1.x * sc.background[2];
                    }
                    flush_color(res, sid, c0.x, c0.y, c1.x);
                    pool.cls[idx] = CLASS_DONE;
                }
                else {
                    int obj = -1, prim = 0;
                    double t = scatter_distance;
                    cls = CLASS_SCATTER; // scatter event before any surface
                    if (found) {
                        t = h.t;
                        obj = h.obj;
                        prim = h.prim;
                        cls = h.cls;
                    }
                    stu4(pool, PL_HIT, idx, static_cast<uint32_t>(__double2loint(t)), static_cast<uint32_t>(__double2hiint(t)),
                         static_cast<uint32_t>(obj), static_cast<uint32_t>(prim));
                    pool.cls[idx] = static_cast<uint8_t>(cls);
                    if (color_changed) {
                        st2(pool, PL_COL0, idx, c0.x, c0.y);
                        st2(pool, PL_COL1, idx, c1.x, c1.y);
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < N_CLASSES; c++)
            if ((sc.class_mask >> c) & 1u) my_counts[c] += static_cast<uint32_t>(__popcll(__ballot(cls == c)));
    }
    for (int off = 32; off > 0; off >>= 1) {
        segments += __shfl_down(segments, off);
        extensions += __shfl_down(extensions, off);
        connections += __shfl_down(connections, off);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int c = 0; c < N_CLASSES; c++) wave_counts[wave][c] = my_counts[c];
        wave_rays[wave][0] = segments;
        wave_rays[wave][1] = extensions;
        wave_rays[wave][2] = connections;
    }
    __syncthreads();
    if (threadIdx.x < N_CLASSES) {
        uint32_t total = 0;
        for (int w = 0; w < BLOCK / 64; w++) total += wave_counts[w][threadIdx.x];
        block_counts[threadIdx.x * count_stride + blockIdx.x] = total;
    }
    if (threadIdx.x >= 64 && threadIdx.x < 67) { // statistics: one atomic per block and counter
        const int k = threadIdx.x - 64;
        uint32_t total = 0;
        for (int w = 0; w < BLOCK / 64; w++) total += wave_rays[w][k];
        if (total) atomicAdd(counters + k, static_cast<unsigned long long>(total));
    }
}

// Exclusive scan, in class-major order, of block_counts[class * count_stride + block] for the first
// n_blocks blocks of every class in class_mask; one workgroup.  Leaves the offsets in place and the
// per-class segment starts, counts and the total in `info`.  Tiles of 4096 counts: one uint4 per
// thread, shuffles inside a wave, 16 wave sums through LDS, a running carry across tiles.
__global__ __launch_bounds__(1024) void k_bin_scan(uint32_t* block_counts, uint32_t count_stride, uint32_t class_mask,
                                                   const BinInfo* previous, BinInfo* info) {
    const uint32_t n_blocks = (previous->total + BIN_CHUNK - 1) / BIN_CHUNK;
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

// Writes the class-sorted index list: record i of class c goes to block_offsets[c][block] + (number
// of class-c records before it in the block).  Records of paths that ended are left out.
__global__ __launch_bounds__(BLOCK) void k_bin_scatter(PathPool pool, const BinInfo* previous, const uint32_t* block_offsets,
                                                       uint32_t count_stride, uint32_t class_mask, uint32_t* list_out) {
    const uint32_t n = previous->total;
    if (blockIdx.x * BIN_CHUNK >= n) return;
    __shared__ uint32_t counts[BIN_ROUNDS][BLOCK / 64][N_CLASSES];
    const int wave = threadIdx.x >> 6;
    int classes[BIN_ROUNDS];
    uint32_t below[BIN_ROUNDS];
#pragma unroll
    for (int r = 0; r < BIN_ROUNDS; r++) {
        const uint32_t idx = blockIdx.x * BIN_CHUNK + r * BLOCK + threadIdx.x;
        int cls = -1;
        if (idx < n) {
            const uint8_t c = pool.cls[idx];
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
    // exclusive prefix over (round, wave) for each class, plus the block's global offset
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
        if (classes[r] >= 0) list_out[counts[r][wave][classes[r]] + below[r]] = blockIdx.x * BIN_CHUNK + r * BLOCK + threadIdx.x;
}

// -------------------------------------------------------------------------------------------------
// Loop body of raytrace (app/pathtracer.cpp:69-171) for the vertices of one class.  Reads record
// list[first + j] of `in`, writes record first + j of `out`.
// counters[3]: records that needed nothing further (path ended, no connection pending).
template<int CLS, bool TRIS, bool UV>
__global__ __launch_bounds__(BLOCK) PTK_WAVES_PER_SIMD(PTK_SHADE_WAVES) void k_shade(DevScene sc, BatchParams b, PathPool in, PathPool out, SampleResults res,
                                                 const uint32_t* list, const BinInfo* info, unsigned long long* counters) {
    const uint32_t n = info->count[CLS];
    const uint32_t first = info->start[CLS];
    uint32_t finished = 0;
    for (uint32_t qi = blockIdx.x * BLOCK + threadIdx.x; qi < n; qi += gridDim.x * BLOCK) {
        const uint32_t dst = first + qi;
        const uint32_t src = list[dst];
        const double2 r0 = ld2(in, PL_RAY0, src), r1 = ld2(in, PL_RAY1, src), r2 = ld2(in, PL_RAY2, src);
        const double2 t0 = ld2(in, PL_THR0, src), t1 = ld2(in, PL_THR1, src);
        const double2 v0 = ld2(in, PL_SAV0, src), v1 = ld2(in, PL_SAV1, src);
        double2 c0 = ld2(in, PL_COL0, src), c1 = ld2(in, PL_COL1, src);
        const uint4 ids = ldu4(in, PL_IDS, src);
        const uint4 hit = ldu4(in, PL_HIT, src);
        V3 org = v3(r0.x, r0.y, r1.x);
        V3 dir = v3(r1.y, r2.x, r2.y);
        V3 thr = v3(t0.x, t0.y, t1.x);
        V3 saved = v3(v0.x, v0.y, v1.x);
        double scatter_pdf = t1.y;
        const uint32_t flags = ids.x;
        const uint32_t sid = ids.y;
        const uint32_t depth = flags & DEPTH_MASK;
        uint32_t ray_type = (flags >> TYPE_SHIFT) & 3u;
        bool has_hit = (flags & FLAG_HAS_HIT) != 0;
        uint32_t medium = flags >> 16;
        if (medium == MEDIUM_NONE) medium = static_cast<uint32_t>(sc.background_medium);
        const DevMedium& med = sc.media[medium];
        uint32_t pixel, sample;
        sample_coordinates(b, sid, pixel, sample);
        Rng rng{rng_key(b.seed, pixel, sample), ids.z};
        if (med.type == PTK_MEDIUM_HOMOGENEOUS_SCATTERING) {
            // the numbers k_trace drew for the free-flight distance: the channel, and the
            // distance only when that channel attenuates (src/medium.cpp:45-51)
            int channel = rng.integer(0, 3);
            if (med.extinction[channel] != 0) rng.ctr++;
        }
        const double t = __hiloint2double(static_cast<int>(hit.y), static_cast<int>(hit.x));
        const V3 point = org + dir * t;
        int next_medium = static_cast<int>(medium);
        V3 n_dir;
        uint32_t sh_info = 0;
        V3 sh_dir = v3(0.0), sh_contrib = v3(0.0);
        double sh_dist = 0;
        uint64_t sh_key = 0;

        if (CLS == CLASS_SCATTER) {
            // :71-93
            V3 emission = medium_emission(med) * thr;
            c0.x += emission.x;
            c0.y += emission.y;
            c1.x += emission.z;
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
                    sh_info = SHADOW_VALID | static_cast<uint32_t>(ls.light_index) | (medium << 16);
                }
                ray_type = RAY_DIFFUSE;
                scatter_pdf = PHASE_ISOTROPIC;
                saved = point;
            }
        }
        else {
            // :94-155
            thr = thr * medium_sample_weight(med, t, false);
            const int obj = static_cast<int>(hit.z);
            const int prim = static_cast<int>(hit.w);
            const DevObject& o = sc.objects[obj];
            const DevMaterial& m = sc.materials[o.material];
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
                V3 add = material_emittance(sc, m, s.use_uv, s.u, s.v) * weight * thr;
                c0.x += add.x;
                c0.y += add.y;
                c1.x += add.z;
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
                            sh_info = SHADOW_VALID | static_cast<uint32_t>(ls.light_index) | (medium << 16);
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

        // Russian roulette :157-171
        bool allow = true;
        if (static_cast<int>(depth) >= sc.min_depth) {
            double threshold = fmin(vmax(thr), 0.9);
            allow = rng.u01() < threshold;
            if (allow) thr = thr / threshold;
        }
        const bool alive = allow && static_cast<int>(depth) + 1 <= sc.max_depth;

        if (!alive && !(sh_info & SHADOW_VALID)) { // nothing left to do for this path
            flush_color(res, sid, c0.x, c0.y, c1.x);
            out.cls[dst] = CLASS_DONE;
            finished++;
            continue;
        }
        // the new ray's origin is also the origin of the pending connection
        st2(out, PL_RAY0, dst, point.x, point.y);
        st2(out, PL_RAY1, dst, point.z, n_dir.x);
        st2(out, PL_SAV1, dst, saved.z, sh_contrib.z);
        st2(out, PL_COL0, dst, c0.x, c0.y);
        st2(out, PL_COL1, dst, c1.x, key_as_double(sh_key));
        stu4(out, PL_IDS, dst,
             (depth + 1) | (ray_type << TYPE_SHIFT) | (has_hit ? FLAG_HAS_HIT : 0u) |
                 ((next_medium < 0 ? MEDIUM_NONE : static_cast<uint32_t>(next_medium)) << 16),
             sid, rng.ctr, sh_info);
        if (sh_info & SHADOW_VALID) {
            st2(out, PL_SH0, dst, sh_dir.x, sh_dir.y);
            st2(out, PL_SH1, dst, sh_dir.z, sh_dist);
            st2(out, PL_SH2, dst, sh_contrib.x, sh_contrib.y);
        }
        if (alive) {
            st2(out, PL_RAY2, dst, n_dir.y, n_dir.z);
            st2(out, PL_THR0, dst, thr.x, thr.y);
            st2(out, PL_THR1, dst, thr.z, scatter_pdf);
            st2(out, PL_SAV0, dst, saved.x, saved.y);
        }
        out.cls[dst] = alive ? 0 : CLASS_DEAD;
    }
    // statistics: one atomic per wave that finished anything (rare: most ended paths still connect)
    for (int off = 32; off > 0; off >>= 1) finished += __shfl_down(finished, off);
    if ((threadIdx.x & 63) == 0 && finished) atomicAdd(counters + 3, static_cast<unsigned long long>(finished));
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

void launch_generate(hipStream_t s, const DevScene& sc, const BatchParams& b, const PathPool& pool, const SampleResults& res,
                     BinInfo* first_info) {
    hipLaunchKernelGGL(k_generate, dim3(grid_for(b.n_paths, 4096)), dim3(BLOCK), 0, s, sc, b, pool, res, first_info);
}

void launch_trace(hipStream_t s, const DevScene& sc, const BatchParams& b, const PathPool& pool, const SampleResults& res, uint32_t bound,
                  const BinInfo* previous, uint32_t* block_counts, uint32_t count_stride, unsigned long long* counters) {
    const dim3 grid(bin_blocks(bound)), block(BLOCK);
#define PTK_LAUNCH_TRACE(T, U) \
    hipLaunchKernelGGL((k_trace<T, U>), grid, block, 0, s, sc, b, pool, res, previous, block_counts, count_stride, counters)
    if (sc.has_triangles) {
        if (sc.needs_uv) PTK_LAUNCH_TRACE(true, true);
        else PTK_LAUNCH_TRACE(true, false);
    }
    else {
        if (sc.needs_uv) PTK_LAUNCH_TRACE(false, true);
        else PTK_LAUNCH_TRACE(false, false);
    }
#undef PTK_LAUNCH_TRACE
}

void launch_bin(hipStream_t s, const DevScene& sc, const PathPool& pool, uint32_t bound, const BinInfo* previous, uint32_t* block_counts,
                uint32_t count_stride, BinInfo* info, uint32_t* list_out) {
    hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(1024), 0, s, block_counts, count_stride, sc.class_mask, previous, info);
    hipLaunchKernelGGL(k_bin_scatter, dim3(bin_blocks(bound)), dim3(BLOCK), 0, s, pool, previous, block_counts, count_stride,
                       sc.class_mask, list_out);
}

template<int CLS>
static void launch_shade_t(hipStream_t s, uint32_t bound, const DevScene& sc, const BatchParams& b, const PathPool& in,
                           const PathPool& out, const SampleResults& res, const uint32_t* list, const BinInfo* info,
                           unsigned long long* counters) {
    const dim3 grid(grid_for(bound, 8192)), block(BLOCK);
#define PTK_LAUNCH_SHADE(T, U) hipLaunchKernelGGL((k_shade<CLS, T, U>), grid, block, 0, s, sc, b, in, out, res, list, info, counters)
    if (sc.has_triangles) {
        if (sc.needs_uv) PTK_LAUNCH_SHADE(true, true);
        else PTK_LAUNCH_SHADE(true, false);
    }
    else {
        if (sc.needs_uv) PTK_LAUNCH_SHADE(false, true);
        else PTK_LAUNCH_SHADE(false, false);
    }
#undef PTK_LAUNCH_SHADE
}

void launch_shade(hipStream_t s, int cls, uint32_t bound, const DevScene& sc, const BatchParams& b, const PathPool& in,
                  const PathPool& out, const SampleResults& res, const uint32_t* list, const BinInfo* info,
                  unsigned long long* counters) {
    switch (cls) {
        case 0: launch_shade_t<0>(s, bound, sc, b, in, out, res, list, info, counters); break;
        case 1: launch_shade_t<1>(s, bound, sc, b, in, out, res, list, info, counters); break;
        case 2: launch_shade_t<2>(s, bound, sc, b, in, out, res, list, info, counters); break;
        case 3: launch_shade_t<3>(s, bound, sc, b, in, out, res, list, info, counters); break;
        case 4: launch_shade_t<4>(s, bound, sc, b, in, out, res, list, info, counters); break;
        case 5: launch_shade_t<5>(s, bound, sc, b, in, out, res, list, info, counters); break;
        case 6: launch_shade_t<6>(s, bound, sc, b, in, out, res, list, info, counters); break;
        case 7: launch_shade_t<7>(s, bound, sc, b, in, out, res, list, info, counters); break;
        default: break;
    }
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
