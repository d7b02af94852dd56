// Wavefront path-tracing kernels for gfx950.
//
//   k_generate     camera samples -> path state              (compute_pixel_color + Camera::make_ray)
//   k_extend       free-flight distance + closest hit; records the material class the path hit and
//                  leaves per-block class counts               (raytrace's loop head + find_closest_hit)
//   k_bin_scan     exclusive scan of the per-block class counts
//   k_bin_scatter  writes the class-sorted list of live paths  (stable: list order is preserved)
//   k_shade<C>     one launch per material class over its segment of the list: emission with MIS,
//                  next-event sampling up to the shadow-ray record, BRDF sampling, medium switch,
//                  Russian roulette; state is updated in place  (raytrace's loop body)
//   k_connect      visibility march of the pending shadow rays  (compute_visibility)
//   k_accumulate   per-pixel sample means folded into the film  (compute_pixel_color's division and
//                  increment_average)
//
// One lane owns one path; a path keeps its slot for life and kernels reach the live paths through an
// index list.  Sorting by class uses block-local ballots and a scan instead of global atomics: with
// one atomic per wave on a shared counter the kernels ran at the rate of that one cache line
// (about 100 atomics per microsecond) instead of at the rate of the arithmetic.  All arithmetic is
// fp64, as in the reference; MFMA has no role here (divergent scalar-per-lane work, no dense
// contraction).
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "shading.h"

namespace ptk {

constexpr int BLOCK = 256;
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

// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_generate(DevScene sc, BatchParams b, PathState st, uint32_t* list, BinInfo* first_info) {
    if (blockIdx.x == 0 && threadIdx.x == 0) first_info->total = b.n_paths; // length of the first list
    for (uint32_t sid = blockIdx.x * BLOCK + threadIdx.x; sid < b.n_paths; sid += gridDim.x * BLOCK) {
        uint32_t pixel, sample;
        sample_coordinates(b, sid, pixel, sample);
        Rng rng{rng_key(b.seed, pixel, sample), 0};
        V3 org, dir;
        camera_ray(sc, pixel, rng, org, dir);
        st.ox[sid] = org.x;
        st.oy[sid] = org.y;
        st.oz[sid] = org.z;
        st.dx[sid] = dir.x;
        st.dy[sid] = dir.y;
        st.dz[sid] = dir.z;
        st.tx[sid] = 1.0;
        st.ty[sid] = 1.0;
        st.tz[sid] = 1.0;
        st.sx[sid] = 0.0;
        st.sy[sid] = 0.0;
        st.sz[sid] = 0.0;
        st.spdf[sid] = 0.0;
        st.ctr[sid] = rng.ctr;
        st.flags[sid] = 0u | (RAY_TRANSMITTED << TYPE_SHIFT) | (static_cast<uint32_t>(sc.background_medium) << 16);
        st.cls[sid] = 0;
        st.sh_info[sid] = 0;
        st.cr[sid] = 0.0;
        st.cg[sid] = 0.0;
        st.cb[sid] = 0.0;
        st.px[sid] = 0.0;
        st.py[sid] = 0.0;
        st.pz[sid] = 0.0;
        st.nx[sid] = 0.0;
        st.ny[sid] = 0.0;
        st.nz[sid] = 0.0;
        list[sid] = sid;
    }
}

// -------------------------------------------------------------------------------------------------
// Loop head of raytrace (app/pathtracer.cpp:53-67): free-flight distance, closest hit, miss handling.
// Block b owns list entries [b * BIN_CHUNK, (b + 1) * BIN_CHUNK) and leaves its per-class counts in
// block_counts[class * count_stride + b].
__global__ __launch_bounds__(BLOCK) void k_extend(DevScene sc, BatchParams b, PathState ps, const uint32_t* list, const BinInfo* previous,
                                                  uint32_t* block_counts, uint32_t count_stride, unsigned long long* ray_counter) {
    const uint32_t n = previous->total; // length of `list`: the sorted list of the previous bounce
    if (blockIdx.x * BIN_CHUNK >= n) return;
    __shared__ StackEntry lds_stack[LDS_STACK_DEPTH * BLOCK];
    __shared__ uint32_t wave_counts[BLOCK / 64][N_CLASSES];
    __shared__ uint32_t wave_rays[BLOCK / 64];
    uint32_t my_rays = 0;
    TraversalStack st = make_stack(lds_stack);
    uint32_t my_counts[N_CLASSES];
    uint32_t wave_rays_acc = 0;
#pragma unroll
    for (int c = 0; c < N_CLASSES; c++) my_counts[c] = 0;
    for (int r = 0; r < BIN_ROUNDS; r++) {
        const uint32_t idx = blockIdx.x * BIN_CHUNK + r * BLOCK + threadIdx.x;
        int cls = -1;
        if (idx < n) {
            const uint32_t slot = list[idx];
            if (ps.cls[slot] != CLASS_DEAD) {
                V3 org = v3(ps.ox[slot], ps.oy[slot], ps.oz[slot]);
                V3 dir = v3(ps.dx[slot], ps.dy[slot], ps.dz[slot]);
                const uint32_t flags = ps.flags[slot];
                uint32_t medium = flags >> 16;
                if (medium == MEDIUM_NONE) medium = static_cast<uint32_t>(sc.background_medium);
                const DevMedium& med = sc.media[medium];
                double scatter_distance = INF;
                if (med.type == PTK_MEDIUM_HOMOGENEOUS_SCATTERING) { // HomogenousScatteringMedium::sample_distance
                    uint32_t pixel, sample;
                    sample_coordinates(b, slot, pixel, sample);
                    Rng rng{rng_key(b.seed, pixel, sample), ps.ctr[slot]};
                    int channel = rng.integer(0, 3);
                    double ext = med.extinction[channel];
                    if (ext != 0) scatter_distance = -log(rng.u01()) / ext;
                }
                HitInfo h;
                bool found = closest_hit(sc, org, dir, scatter_distance, h, st);
                my_rays = 1;
                if (!found && scatter_distance == INF) {
                    // color += throughput * background_color; the path ends
                    if (!sc.background_is_black) {
                        ps.cr[slot] += ps.tx[slot] * sc.background[0];
                        ps.cg[slot] += ps.ty[slot] * sc.background[1];
                        ps.cb[slot] += ps.tz[slot] * sc.background[2];
                    }
                    ps.cls[slot] = CLASS_DEAD;
                }
                else if (!found) { // scatter event before any surface
                    ps.hit_t[slot] = scatter_distance;
                    ps.hit_obj[slot] = -1;
                    ps.hit_prim[slot] = 0;
                    cls = CLASS_SCATTER;
                    ps.cls[slot] = CLASS_SCATTER;
                }
                else {
                    ps.hit_t[slot] = h.t;
                    ps.hit_obj[slot] = h.obj;
                    ps.hit_prim[slot] = h.prim;
                    cls = sc.materials[sc.objects[h.obj].material].type;
                    ps.cls[slot] = static_cast<uint8_t>(cls);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < N_CLASSES; c++)
            if ((sc.class_mask >> c) & 1u) my_counts[c] += static_cast<uint32_t>(__popcll(__ballot(cls == c)));
        wave_rays_acc += static_cast<uint32_t>(__popcll(__ballot(my_rays != 0)));
        my_rays = 0;
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int c = 0; c < N_CLASSES; c++) wave_counts[wave][c] = my_counts[c];
        wave_rays[wave] = wave_rays_acc;
    }
    __syncthreads();
    if (threadIdx.x < N_CLASSES) {
        uint32_t total = 0;
        for (int w = 0; w < BLOCK / 64; w++) total += wave_counts[w][threadIdx.x];
        block_counts[threadIdx.x * count_stride + blockIdx.x] = total;
    }
    if (threadIdx.x == 0) { // rays traced, for the statistics: one atomic per block
        uint32_t total = 0;
        for (int w = 0; w < BLOCK / 64; w++) total += wave_rays[w];
        if (total) atomicAdd(ray_counter, static_cast<unsigned long long>(total));
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

// Writes the class-sorted list: entry i of the input list with class c goes to
// block_offsets[c][block] + (number of class-c entries before it in the block).  Paths that ended
// (CLASS_DEAD) are dropped.
__global__ __launch_bounds__(BLOCK) void k_bin_scatter(PathState ps, const uint32_t* list_in, const BinInfo* previous,
                                                       const uint32_t* block_offsets, uint32_t count_stride, uint32_t class_mask,
                                                       uint32_t* list_out) {
    const uint32_t n = previous->total;
    if (blockIdx.x * BIN_CHUNK >= n) return;
    __shared__ uint32_t counts[BIN_ROUNDS][BLOCK / 64][N_CLASSES];
    const int wave = threadIdx.x >> 6;
    uint32_t slots[BIN_ROUNDS];
    int classes[BIN_ROUNDS];
    uint32_t below[BIN_ROUNDS];
#pragma unroll
    for (int r = 0; r < BIN_ROUNDS; r++) {
        const uint32_t idx = blockIdx.x * BIN_CHUNK + r * BLOCK + threadIdx.x;
        int cls = -1;
        uint32_t slot = 0;
        if (idx < n) {
            slot = list_in[idx];
            const uint8_t c = ps.cls[slot];
            if (c != CLASS_DEAD) cls = c;
        }
        slots[r] = slot;
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
        if (classes[r] >= 0) list_out[counts[r][wave][classes[r]] + below[r]] = slots[r];
}

// -------------------------------------------------------------------------------------------------
// Loop body of raytrace (app/pathtracer.cpp:69-171) for the paths of one class.
template<int CLS>
__global__ __launch_bounds__(BLOCK) void k_shade(DevScene sc, BatchParams b, PathState ps, const uint32_t* list, const BinInfo* info) {
    const uint32_t n = info->count[CLS];
    const uint32_t first = info->start[CLS];
    for (uint32_t qi = blockIdx.x * BLOCK + threadIdx.x; qi < n; qi += gridDim.x * BLOCK) {
        const uint32_t slot = list[first + qi];
        V3 org = v3(ps.ox[slot], ps.oy[slot], ps.oz[slot]);
        V3 dir = v3(ps.dx[slot], ps.dy[slot], ps.dz[slot]);
        V3 thr = v3(ps.tx[slot], ps.ty[slot], ps.tz[slot]);
        V3 saved = v3(ps.sx[slot], ps.sy[slot], ps.sz[slot]);
        double scatter_pdf = ps.spdf[slot];
        const uint32_t flags = ps.flags[slot];
        const uint32_t depth = flags & DEPTH_MASK;
        uint32_t ray_type = (flags >> TYPE_SHIFT) & 3u;
        bool has_hit = (flags & FLAG_HAS_HIT) != 0;
        uint32_t medium = flags >> 16;
        if (medium == MEDIUM_NONE) medium = static_cast<uint32_t>(sc.background_medium);
        const DevMedium& med = sc.media[medium];
        uint32_t pixel, sample;
        sample_coordinates(b, slot, pixel, sample);
        Rng rng{rng_key(b.seed, pixel, sample), ps.ctr[slot]};
        if (med.type == PTK_MEDIUM_HOMOGENEOUS_SCATTERING) {
            // the numbers k_extend drew for the free-flight distance: the channel, and the
            // distance only when that channel attenuates (src/medium.cpp:45-51)
            int channel = rng.integer(0, 3);
            if (med.extinction[channel] != 0) rng.ctr++;
        }
        const double t = ps.hit_t[slot];
        const V3 point = org + dir * t;
        int next_medium = static_cast<int>(medium);
        V3 n_dir;
        uint32_t sh_info = 0;

        if (CLS == CLASS_SCATTER) {
            // :71-93
            V3 emission = medium_emission(med) * thr;
            if (emission.x != 0 || emission.y != 0 || emission.z != 0) {
                ps.cr[slot] += emission.x;
                ps.cg[slot] += emission.y;
                ps.cb[slot] += emission.z;
            }
            thr = thr * medium_sample_weight(med, t, true);
            n_dir = sample_spherical(rng);
            if (sc.nee) {
                LightSample ls;
                if (pick_light_point(sc, -1, point, ls, rng)) {
                    double weight = mis_weight(ls.light_pdf, PHASE_ISOTROPIC);
                    V3 contrib = thr * (weight * PHASE_ISOTROPIC / ls.light_pdf * static_cast<double>(sc.n_lights));
                    ps.sh_dx[slot] = ls.direction.x;
                    ps.sh_dy[slot] = ls.direction.y;
                    ps.sh_dz[slot] = ls.direction.z;
                    ps.sh_dist[slot] = ls.distance;
                    ps.sh_cr[slot] = contrib.x;
                    ps.sh_cg[slot] = contrib.y;
                    ps.sh_cb[slot] = contrib.z;
                    ps.sh_key[slot] = rng_fork(rng.key, rng.ctr);
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
            const int obj = ps.hit_obj[slot];
            const int prim = ps.hit_prim[slot];
            const DevObject& o = sc.objects[obj];
            const DevMaterial& m = sc.materials[o.material];
            Surface s;
            s.point = point;
            s.incident = dir;
            s.obj = obj;
            s.prim = prim;
            V3 normal = object_normal(sc, o, point, prim);
            s.outside = dot(dir, normal) < 0;
            s.normal = s.outside ? normal : -normal;
            s.use_uv = m.needs_uv != 0;
            s.u = s.v = 0;
            if (s.use_uv) object_uv(sc, o, point, prim, s.u, s.v);

            if (!has_hit) {
                ps.px[slot] = point.x;
                ps.py[slot] = point.y;
                ps.pz[slot] = point.z;
                ps.nx[slot] = s.normal.x;
                ps.ny[slot] = s.normal.y;
                ps.nz[slot] = s.normal.z;
                has_hit = true;
            }
            const bool is_specular = ray_type == RAY_REFLECTED || ray_type == RAY_TRANSMITTED;
            if (m.is_light) { // :106-119
                double weight = 1;
                if (sc.nee && depth != 0 && !is_specular)
                    weight = mis_weight(scatter_pdf, object_light_pdf(sc, o, point, saved, prim));
                V3 add = material_emittance(sc, m, s.use_uv, s.u, s.v) * weight * thr;
                ps.cr[slot] += add.x;
                ps.cg[slot] += add.y;
                ps.cb[slot] += add.z;
            }
            if (sc.nee) { // sample_light up to the march (src/objects.cpp:515-551, 562-578)
                LightSample ls;
                if (pick_light_point(sc, obj, point, ls, rng)) {
                    V3 brdf = material_eval<CLS>(sc, m, s, ls.direction, rng);
                    if (len2(brdf) != 0) {
                        double pdf = material_pdf<CLS>(sc, m, s, ls.direction);
                        const uint64_t key = rng_fork(rng.key, rng.ctr);
                        rng.ctr++;
                        bool wrong_side = dot(s.incident, s.normal) * dot(ls.direction, s.normal) > 0;
                        double cosine = fmax(dot(s.normal, ls.direction), 0.0);
                        if (!wrong_side && cosine > 0) {
                            double weight = mis_weight(ls.light_pdf, pdf);
                            V3 contrib = brdf * thr * (weight * cosine / ls.light_pdf * static_cast<double>(sc.n_lights));
                            ps.sh_dx[slot] = ls.direction.x;
                            ps.sh_dy[slot] = ls.direction.y;
                            ps.sh_dz[slot] = ls.direction.z;
                            ps.sh_dist[slot] = ls.distance;
                            ps.sh_cr[slot] = contrib.x;
                            ps.sh_cg[slot] = contrib.y;
                            ps.sh_cb[slot] = contrib.z;
                            ps.sh_key[slot] = key;
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

        ps.sh_info[slot] = sh_info;
        ps.ox[slot] = point.x; // the new ray's origin, and the origin of the pending shadow ray
        ps.oy[slot] = point.y;
        ps.oz[slot] = point.z;
        if (alive) {
            ps.dx[slot] = n_dir.x;
            ps.dy[slot] = n_dir.y;
            ps.dz[slot] = n_dir.z;
            ps.tx[slot] = thr.x;
            ps.ty[slot] = thr.y;
            ps.tz[slot] = thr.z;
            ps.sx[slot] = saved.x;
            ps.sy[slot] = saved.y;
            ps.sz[slot] = saved.z;
            ps.spdf[slot] = scatter_pdf;
            ps.ctr[slot] = rng.ctr;
            ps.flags[slot] = (depth + 1) | (ray_type << TYPE_SHIFT) | (has_hit ? FLAG_HAS_HIT : 0u) |
                             ((next_medium < 0 ? MEDIUM_NONE : static_cast<uint32_t>(next_medium)) << 16);
        }
        else {
            ps.cls[slot] = CLASS_DEAD;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Visibility march for every path of the bounce's list that has a pending connection.
__global__ __launch_bounds__(BLOCK) void k_connect(DevScene sc, PathState ps, const uint32_t* list, const BinInfo* info,
                                                   unsigned long long* ray_counter) {
    __shared__ StackEntry lds_stack[LDS_STACK_DEPTH * BLOCK];
    __shared__ uint32_t block_rays;
    TraversalStack st = make_stack(lds_stack);
    if (threadIdx.x == 0) block_rays = 0;
    __syncthreads();
    const uint32_t n = info->total;
    uint32_t rays = 0;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const uint32_t slot = list[i];
        const uint32_t sh = ps.sh_info[slot];
        if (!(sh & SHADOW_VALID)) continue;
        V3 org = v3(ps.ox[slot], ps.oy[slot], ps.oz[slot]);
        V3 dir = v3(ps.sh_dx[slot], ps.sh_dy[slot], ps.sh_dz[slot]);
        uint32_t medium = (sh >> 16) & 0x7FFFu;
        Rng rng{ps.sh_key[slot], 0};
        V3 visible = march_to_light(sc, org, dir, static_cast<int>(medium), static_cast<int>(sh & 0xFFFFu), ps.sh_dist[slot], rng,
                                    st, rays);
        if (visible.x != 0 || visible.y != 0 || visible.z != 0) {
            ps.cr[slot] += ps.sh_cr[slot] * visible.x;
            ps.cg[slot] += ps.sh_cg[slot] * visible.y;
            ps.cb[slot] += ps.sh_cb[slot] * visible.z;
        }
    }
    // shadow-ray count for the statistics: one global atomic per block
    for (int off = 32; off > 0; off >>= 1) rays += __shfl_down(rays, off);
    if ((threadIdx.x & 63) == 0 && rays) atomicAdd(&block_rays, rays);
    __syncthreads();
    if (threadIdx.x == 0 && block_rays) atomicAdd(ray_counter, static_cast<unsigned long long>(block_rays));
}

// -------------------------------------------------------------------------------------------------
// compute_pixel_color's mean over the batch's samples (app/pathtracer.cpp:192-199) and
// raytrace_section's running mean (:232-235, 253-267); one lane per pixel, samples in order.
__global__ __launch_bounds__(BLOCK) void k_accumulate(BatchParams b, PathState res, double* image, double* position,
                                                      double* normal, unsigned long long* pixel_counters) {
    for (uint32_t p = blockIdx.x * BLOCK + threadIdx.x; p < b.n_pixels; p += gridDim.x * BLOCK) {
        double sum[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (uint32_t s = 0; s < b.n_samples; s++) {
            const size_t sid = static_cast<size_t>(s) * b.n_pixels + p;
            sum[0] += res.cr[sid];
            sum[1] += res.cg[sid];
            sum[2] += res.cb[sid];
            sum[3] += res.px[sid];
            sum[4] += res.py[sid];
            sum[5] += res.pz[sid];
            sum[6] += res.nx[sid];
            sum[7] += res.ny[sid];
            sum[8] += res.nz[sid];
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
__global__ __launch_bounds__(BLOCK) void k_gather_samples(BatchParams b, PathState res, size_t pixel_base,
                                                          double* color, double* position, double* normal) {
    for (uint32_t sid = blockIdx.x * BLOCK + threadIdx.x; sid < b.n_paths; sid += gridDim.x * BLOCK) {
        uint32_t s = sid / b.n_pixels, p = sid - s * b.n_pixels;
        size_t o = 3 * ((pixel_base + p) * b.n_samples + s);
        if (color) {
            color[o] = res.cr[sid];
            color[o + 1] = res.cg[sid];
            color[o + 2] = res.cb[sid];
        }
        if (position) {
            position[o] = res.px[sid];
            position[o + 1] = res.py[sid];
            position[o + 2] = res.pz[sid];
        }
        if (normal) {
            normal[o] = res.nx[sid];
            normal[o + 1] = res.ny[sid];
            normal[o + 2] = res.nz[sid];
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
        bool found = closest_hit(sc, org, dir, INF, h, st);
        V3 point = v3(0.0), n = v3(0.0);
        int prim = 0;
        if (found) {
            const DevObject& o = sc.objects[h.obj];
            point = org + dir * h.t;
            n = object_normal(sc, o, point, h.prim);
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

void launch_generate(hipStream_t s, const DevScene& sc, const BatchParams& b, const PathState& ps, uint32_t* list, BinInfo* first_info) {
    hipLaunchKernelGGL(k_generate, dim3(grid_for(b.n_paths, 4096)), dim3(BLOCK), 0, s, sc, b, ps, list, first_info);
}

void launch_extend(hipStream_t s, const DevScene& sc, const BatchParams& b, const PathState& ps, const uint32_t* list, uint32_t bound,
                   const BinInfo* previous, uint32_t* block_counts, uint32_t count_stride, unsigned long long* ray_counter) {
    hipLaunchKernelGGL(k_extend, dim3(bin_blocks(bound)), dim3(BLOCK), 0, s, sc, b, ps, list, previous, block_counts, count_stride,
                       ray_counter);
}

void launch_bin(hipStream_t s, const DevScene& sc, const PathState& ps, const uint32_t* list_in, uint32_t bound, const BinInfo* previous,
                uint32_t* block_counts, uint32_t count_stride, BinInfo* info, uint32_t* list_out) {
    hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(1024), 0, s, block_counts, count_stride, sc.class_mask, previous, info);
    hipLaunchKernelGGL(k_bin_scatter, dim3(bin_blocks(bound)), dim3(BLOCK), 0, s, ps, list_in, previous, block_counts, count_stride,
                       sc.class_mask, list_out);
}

template<int CLS>
static void launch_shade_t(hipStream_t s, uint32_t bound, const DevScene& sc, const BatchParams& b, const PathState& ps,
                           const uint32_t* list, const BinInfo* info) {
    hipLaunchKernelGGL(k_shade<CLS>, dim3(grid_for(bound, 16384)), dim3(BLOCK), 0, s, sc, b, ps, list, info);
}

void launch_shade(hipStream_t s, int cls, uint32_t bound, const DevScene& sc, const BatchParams& b, const PathState& ps,
                  const uint32_t* list, const BinInfo* info) {
    switch (cls) {
        case 0: launch_shade_t<0>(s, bound, sc, b, ps, list, info); break;
        case 1: launch_shade_t<1>(s, bound, sc, b, ps, list, info); break;
        case 2: launch_shade_t<2>(s, bound, sc, b, ps, list, info); break;
        case 3: launch_shade_t<3>(s, bound, sc, b, ps, list, info); break;
        case 4: launch_shade_t<4>(s, bound, sc, b, ps, list, info); break;
        case 5: launch_shade_t<5>(s, bound, sc, b, ps, list, info); break;
        case 6: launch_shade_t<6>(s, bound, sc, b, ps, list, info); break;
        case 7: launch_shade_t<7>(s, bound, sc, b, ps, list, info); break;
        default: break;
    }
}

void launch_connect(hipStream_t s, uint32_t bound, const DevScene& sc, const PathState& ps, const uint32_t* list, const BinInfo* info,
                    unsigned long long* ray_counter) {
    hipLaunchKernelGGL(k_connect, dim3(grid_for(bound, 16384)), dim3(BLOCK), 0, s, sc, ps, list, info, ray_counter);
}

void launch_accumulate(hipStream_t s, const BatchParams& b, const PathState& ps, double* image, double* position, double* normal,
                       unsigned long long* pixel_counters) {
    hipLaunchKernelGGL(k_accumulate, dim3(grid_for(b.n_pixels, 4096)), dim3(BLOCK), 0, s, b, ps, image, position, normal,
                       pixel_counters);
}

void launch_gather_samples(hipStream_t s, const BatchParams& b, const PathState& ps, size_t pixel_base, double* color,
                           double* position, double* normal) {
    hipLaunchKernelGGL(k_gather_samples, dim3(grid_for(b.n_paths, 4096)), dim3(BLOCK), 0, s, b, ps, pixel_base, color, position,
                       normal);
}

void launch_primary(hipStream_t s, const DevScene& sc, const BatchParams& b, int32_t* object_index, int32_t* primitive_index,
                    double* distance, double* position, double* normal) {
    hipLaunchKernelGGL(k_primary, dim3(grid_for(b.n_pixels, 4096)), dim3(BLOCK), 0, s, sc, b, object_index, primitive_index,
                       distance, position, normal);
}

} // namespace ptk
