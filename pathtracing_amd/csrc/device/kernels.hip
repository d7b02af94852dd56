// Wavefront path-tracing kernels for gfx950.
//
//   k_generate   camera samples -> path state            (compute_pixel_color + Camera::make_ray)
//   k_extend     free-flight distance + closest hit, then sorts the path into the queue of the
//                material class it hit                    (raytrace's loop head + find_closest_hit)
//   k_shade<C>   one launch per material class: emission with MIS, next-event sampling up to the
//                shadow-ray record, BRDF sampling, medium switch, Russian roulette; survivors are
//                written densely into the other pool       (raytrace's loop body)
//   k_connect    visibility march of the queued shadow rays (compute_visibility)
//   k_accumulate per-pixel sample means folded into the film (compute_pixel_color's division and
//                increment_average)
//
// One lane owns one path.  Queues and the next pool are filled with one atomic per wave:
// ballot -> mbcnt prefix -> lane 0 reserves the range.  All arithmetic is fp64, as in the
// reference; MFMA has no role here (divergent scalar-per-lane work, no dense contraction).
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "shading.h"

namespace ptk {

constexpr int BLOCK = 256;

// ---- wave-level append ------------------------------------------------------------------------
// Reserves one slot for every lane with `active` set; returns the lane's slot (undefined when not
// active).  One atomic per wave.
PTK_DEV uint32_t wave_append(uint32_t* counter, bool active) {
    const unsigned long long mask = __ballot(active);
    const uint32_t total = static_cast<uint32_t>(__popcll(mask));
    if (total == 0) return 0;
    const uint32_t lane_prefix = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32),
                                                           __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0));
    uint32_t base = 0;
    const int leader = __ffsll(static_cast<long long>(mask)) - 1;
    if (static_cast<int>(__lane_id()) == leader) base = atomicAdd(counter, total);
    base = __shfl(base, leader);
    return base + lane_prefix;
}

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
__global__ __launch_bounds__(BLOCK) void k_generate(DevScene sc, BatchParams b, PathPool pool, SampleResults res) {
    for (uint32_t sid = blockIdx.x * BLOCK + threadIdx.x; sid < b.n_paths; sid += gridDim.x * BLOCK) {
        uint32_t pixel, sample;
        sample_coordinates(b, sid, pixel, sample);
        Rng rng{rng_key(b.seed, pixel, sample), 0};
        V3 org, dir;
        camera_ray(sc, pixel, rng, org, dir);
        pool.ox[sid] = org.x;
        pool.oy[sid] = org.y;
        pool.oz[sid] = org.z;
        pool.dx[sid] = dir.x;
        pool.dy[sid] = dir.y;
        pool.dz[sid] = dir.z;
        pool.tx[sid] = 1.0;
        pool.ty[sid] = 1.0;
        pool.tz[sid] = 1.0;
        pool.sx[sid] = 0.0;
        pool.sy[sid] = 0.0;
        pool.sz[sid] = 0.0;
        pool.spdf[sid] = 0.0;
        pool.sid[sid] = sid;
        pool.ctr[sid] = rng.ctr;
        pool.flags[sid] = 0u | (RAY_TRANSMITTED << TYPE_SHIFT) | (static_cast<uint32_t>(sc.background_medium) << 16);
        res.cr[sid] = 0.0;
        res.cg[sid] = 0.0;
        res.cb[sid] = 0.0;
        res.px[sid] = 0.0;
        res.py[sid] = 0.0;
        res.pz[sid] = 0.0;
        res.nx[sid] = 0.0;
        res.ny[sid] = 0.0;
        res.nz[sid] = 0.0;
    }
}

// -------------------------------------------------------------------------------------------------
// Loop head of raytrace (app/pathtracer.cpp:53-67): free-flight distance, closest hit, miss handling.
__global__ __launch_bounds__(BLOCK) void k_extend(DevScene sc, BatchParams b, PathPool pool, uint32_t n_active, HitRecords hits,
                                                  uint32_t* queues, uint32_t queue_stride, BounceCounters* counters,
                                                  SampleResults res) {
    __shared__ StackEntry lds_stack[LDS_STACK_DEPTH * BLOCK];
    TraversalStack st = make_stack(lds_stack);
    const uint32_t rounds = (n_active + gridDim.x * BLOCK - 1) / (gridDim.x * BLOCK);
    for (uint32_t r = 0; r < rounds; r++) { // every lane runs every round: the appends are wave-wide
        const uint32_t slot = (r * gridDim.x + blockIdx.x) * BLOCK + threadIdx.x;
        const bool valid = slot < n_active;
        int cls = -1;
        if (valid) {
            V3 org = v3(pool.ox[slot], pool.oy[slot], pool.oz[slot]);
            V3 dir = v3(pool.dx[slot], pool.dy[slot], pool.dz[slot]);
            const uint32_t flags = pool.flags[slot];
            uint32_t medium = flags >> 16;
            if (medium == MEDIUM_NONE) medium = static_cast<uint32_t>(sc.background_medium);
            const DevMedium& med = sc.media[medium];
            double scatter_distance = INF;
            if (med.type == PTK_MEDIUM_HOMOGENEOUS_SCATTERING) { // HomogenousScatteringMedium::sample_distance
                uint32_t pixel, sample;
                const uint32_t sid = pool.sid[slot];
                sample_coordinates(b, sid, pixel, sample);
                Rng rng{rng_key(b.seed, pixel, sample), pool.ctr[slot]};
                int channel = rng.integer(0, 3);
                double ext = med.extinction[channel];
                if (ext != 0) scatter_distance = -log(rng.u01()) / ext;
            }
            HitInfo h;
            bool found = closest_hit(sc, org, dir, scatter_distance, h, st);
            if (!found && scatter_distance == INF) {
                // color += throughput * background_color; path ends
                if (!sc.background_is_black) {
                    const uint32_t sid = pool.sid[slot];
                    res.cr[sid] += pool.tx[slot] * sc.background[0];
                    res.cg[sid] += pool.ty[slot] * sc.background[1];
                    res.cb[sid] += pool.tz[slot] * sc.background[2];
                }
            }
            else if (!found) { // scatter event before any surface
                hits.t[slot] = scatter_distance;
                hits.obj[slot] = -1;
                hits.prim[slot] = 0;
                cls = CLASS_SCATTER;
            }
            else {
                hits.t[slot] = h.t;
                hits.obj[slot] = h.obj;
                hits.prim[slot] = h.prim;
                cls = sc.materials[sc.objects[h.obj].material].type;
            }
        }
        // sort into per-class queues; classes absent from the scene are skipped
#pragma unroll
        for (int c = 0; c < N_CLASSES; c++) {
            if (!((sc.class_mask >> c) & 1u)) continue;
            const bool mine = cls == c;
            const uint32_t pos = wave_append(&counters->queue_count[c], mine);
            if (mine) queues[static_cast<size_t>(c) * queue_stride + pos] = slot;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Loop body of raytrace (app/pathtracer.cpp:69-171) for the paths of one class.
template<int CLS>
__global__ __launch_bounds__(BLOCK) void k_shade(DevScene sc, BatchParams b, PathPool in, PathPool out, HitRecords hits,
                                                 const uint32_t* queue, BounceCounters* counters, ShadowQueue shadow,
                                                 SampleResults res) {
    const uint32_t n = counters->queue_count[CLS];
    const uint32_t rounds = (n + gridDim.x * BLOCK - 1) / (gridDim.x * BLOCK);
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t qi = (r * gridDim.x + blockIdx.x) * BLOCK + threadIdx.x;
        const bool valid = qi < n;
        bool alive = false, want_shadow = false;
        // next state
        V3 n_org{}, n_dir{}, thr{}, saved{};
        double scatter_pdf = 0;
        uint32_t sid = 0, n_flags = 0;
        Rng rng{0, 0};
        // shadow record
        V3 sh_dir{}, sh_contrib{};
        double sh_dist = 0;
        uint64_t sh_key = 0;
        uint32_t sh_info = 0;

        if (valid) {
            const uint32_t slot = queue[qi];
            V3 org = v3(in.ox[slot], in.oy[slot], in.oz[slot]);
            V3 dir = v3(in.dx[slot], in.dy[slot], in.dz[slot]);
            thr = v3(in.tx[slot], in.ty[slot], in.tz[slot]);
            saved = v3(in.sx[slot], in.sy[slot], in.sz[slot]);
            scatter_pdf = in.spdf[slot];
            sid = in.sid[slot];
            const uint32_t flags = in.flags[slot];
            const uint32_t depth = flags & DEPTH_MASK;
            uint32_t ray_type = (flags >> TYPE_SHIFT) & 3u;
            bool has_hit = (flags & FLAG_HAS_HIT) != 0;
            uint32_t medium = flags >> 16;
            if (medium == MEDIUM_NONE) medium = static_cast<uint32_t>(sc.background_medium);
            const DevMedium& med = sc.media[medium];
            uint32_t pixel, sample;
            sample_coordinates(b, sid, pixel, sample);
            rng.key = rng_key(b.seed, pixel, sample);
            rng.ctr = in.ctr[slot];
            if (med.type == PTK_MEDIUM_HOMOGENEOUS_SCATTERING) {
                // the numbers k_extend drew for the free-flight distance: the channel, and the
                // distance only when that channel attenuates (src/medium.cpp:45-51)
                int channel = rng.integer(0, 3);
                if (med.extinction[channel] != 0) rng.ctr++;
            }

            const double t = hits.t[slot];
            const V3 point = org + dir * t;
            int next_medium = static_cast<int>(medium);

            if (CLS == CLASS_SCATTER) {
                // :71-93
                V3 emission = medium_emission(med) * thr;
                if (emission.x != 0 || emission.y != 0 || emission.z != 0) {
                    res.cr[sid] += emission.x;
                    res.cg[sid] += emission.y;
                    res.cb[sid] += emission.z;
                }
                thr = thr * medium_sample_weight(med, t, true);
                V3 scattered = sample_spherical(rng);
                if (sc.nee) {
                    LightSample ls;
                    if (pick_light_point(sc, -1, point, ls, rng)) {
                        double weight = mis_weight(ls.light_pdf, PHASE_ISOTROPIC);
                        sh_contrib = thr * (weight * PHASE_ISOTROPIC / ls.light_pdf * static_cast<double>(sc.n_lights));
                        sh_dir = ls.direction;
                        sh_dist = ls.distance;
                        sh_key = rng_fork(rng.key, rng.ctr);
                        rng.ctr++;
                        sh_info = static_cast<uint32_t>(ls.light_index) | (medium << 16);
                        want_shadow = true;
                    }
                    ray_type = RAY_DIFFUSE;
                    scatter_pdf = PHASE_ISOTROPIC;
                    saved = point;
                }
                n_org = point;
                n_dir = scattered;
            }
            else {
                // :94-155
                thr = thr * medium_sample_weight(med, t, false);
                const int obj = hits.obj[slot];
                const int prim = hits.prim[slot];
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
                    res.px[sid] = point.x;
                    res.py[sid] = point.y;
                    res.pz[sid] = point.z;
                    res.nx[sid] = s.normal.x;
                    res.ny[sid] = s.normal.y;
                    res.nz[sid] = s.normal.z;
                    has_hit = true;
                }
                const bool is_specular = ray_type == RAY_REFLECTED || ray_type == RAY_TRANSMITTED;
                if (m.is_light) { // :106-119
                    double weight = 1;
                    if (sc.nee && depth != 0 && !is_specular)
                        weight = mis_weight(scatter_pdf, object_light_pdf(sc, o, point, saved, prim));
                    V3 add = material_emittance(sc, m, s.use_uv, s.u, s.v) * weight * thr;
                    res.cr[sid] += add.x;
                    res.cg[sid] += add.y;
                    res.cb[sid] += add.z;
                }
                if (sc.nee) { // sample_light up to the march (src/objects.cpp:515-551, 562-578)
                    LightSample ls;
                    if (pick_light_point(sc, obj, point, ls, rng)) {
                        V3 brdf = material_eval<CLS>(sc, m, s, ls.direction, rng);
                        if (len2(brdf) != 0) {
                            double pdf = material_pdf<CLS>(sc, m, s, ls.direction);
                            sh_key = rng_fork(rng.key, rng.ctr);
                            rng.ctr++;
                            bool wrong_side = dot(s.incident, s.normal) * dot(ls.direction, s.normal) > 0;
                            double cosine = fmax(dot(s.normal, ls.direction), 0.0);
                            if (!wrong_side && cosine > 0) {
                                double weight = mis_weight(ls.light_pdf, pdf);
                                sh_contrib = brdf * thr * (weight * cosine / ls.light_pdf * static_cast<double>(sc.n_lights));
                                sh_dir = ls.direction;
                                sh_dist = ls.distance;
                                sh_info = static_cast<uint32_t>(ls.light_index) | (medium << 16);
                                want_shadow = true;
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
                n_org = point;
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
            alive = allow && static_cast<int>(depth) + 1 <= sc.max_depth;
            n_flags = (depth + 1) | (ray_type << TYPE_SHIFT) | (has_hit ? FLAG_HAS_HIT : 0u) |
                      ((next_medium < 0 ? MEDIUM_NONE : static_cast<uint32_t>(next_medium)) << 16);
        }

        const uint32_t shadow_pos = wave_append(&counters->n_shadow, want_shadow);
        if (want_shadow) {
            shadow.ox[shadow_pos] = n_org.x;
            shadow.oy[shadow_pos] = n_org.y;
            shadow.oz[shadow_pos] = n_org.z;
            shadow.dx[shadow_pos] = sh_dir.x;
            shadow.dy[shadow_pos] = sh_dir.y;
            shadow.dz[shadow_pos] = sh_dir.z;
            shadow.dist[shadow_pos] = sh_dist;
            shadow.cr[shadow_pos] = sh_contrib.x;
            shadow.cg[shadow_pos] = sh_contrib.y;
            shadow.cb[shadow_pos] = sh_contrib.z;
            shadow.key[shadow_pos] = sh_key;
            shadow.sid[shadow_pos] = sid;
            shadow.info[shadow_pos] = sh_info;
        }
        const uint32_t next = wave_append(&counters->n_next, alive);
        if (alive) {
            out.ox[next] = n_org.x;
            out.oy[next] = n_org.y;
            out.oz[next] = n_org.z;
            out.dx[next] = n_dir.x;
            out.dy[next] = n_dir.y;
            out.dz[next] = n_dir.z;
            out.tx[next] = thr.x;
            out.ty[next] = thr.y;
            out.tz[next] = thr.z;
            out.sx[next] = saved.x;
            out.sy[next] = saved.y;
            out.sz[next] = saved.z;
            out.spdf[next] = scatter_pdf;
            out.sid[next] = sid;
            out.ctr[next] = rng.ctr;
            out.flags[next] = n_flags;
        }
    }
}

// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_connect(DevScene sc, ShadowQueue shadow, const BounceCounters* counters,
                                                   SampleResults res, unsigned long long* ray_counter) {
    __shared__ StackEntry lds_stack[LDS_STACK_DEPTH * BLOCK];
    TraversalStack st = make_stack(lds_stack);
    const uint32_t n = counters->n_shadow;
    uint32_t rays = 0;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        V3 org = v3(shadow.ox[i], shadow.oy[i], shadow.oz[i]);
        V3 dir = v3(shadow.dx[i], shadow.dy[i], shadow.dz[i]);
        const uint32_t info = shadow.info[i];
        uint32_t medium = info >> 16;
        Rng rng{shadow.key[i], 0};
        V3 visible = march_to_light(sc, org, dir, static_cast<int>(medium), static_cast<int>(info & 0xFFFFu), shadow.dist[i],
                                    rng, st, rays);
        if (visible.x != 0 || visible.y != 0 || visible.z != 0) {
            const uint32_t sid = shadow.sid[i];
            res.cr[sid] += shadow.cr[i] * visible.x;
            res.cg[sid] += shadow.cg[i] * visible.y;
            res.cb[sid] += shadow.cb[i] * visible.z;
        }
    }
    // shadow-ray count for the statistics: one atomic per wave
    for (int off = 32; off > 0; off >>= 1) rays += __shfl_down(rays, off);
    if (__lane_id() == 0 && rays) atomicAdd(ray_counter, static_cast<unsigned long long>(rays));
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
__global__ __launch_bounds__(BLOCK) void k_gather_samples(BatchParams b, SampleResults res, size_t pixel_base,
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

void launch_generate(hipStream_t s, const DevScene& sc, const BatchParams& b, const PathPool& pool, const SampleResults& res) {
    hipLaunchKernelGGL(k_generate, dim3(grid_for(b.n_paths, 4096)), dim3(BLOCK), 0, s, sc, b, pool, res);
}

void launch_extend(hipStream_t s, const DevScene& sc, const BatchParams& b, const PathPool& pool, uint32_t n_active,
                   const HitRecords& hits, uint32_t* queues, uint32_t queue_stride, BounceCounters* counters,
                   const SampleResults& res) {
    hipLaunchKernelGGL(k_extend, dim3(grid_for(n_active, 8192)), dim3(BLOCK), 0, s, sc, b, pool, n_active, hits, queues,
                       queue_stride, counters, res);
}

template<int CLS>
static void launch_shade_t(hipStream_t s, uint32_t bound, const DevScene& sc, const BatchParams& b, const PathPool& in,
                           const PathPool& out, const HitRecords& hits, const uint32_t* queue, BounceCounters* counters,
                           const ShadowQueue& shadow, const SampleResults& res) {
    hipLaunchKernelGGL(k_shade<CLS>, dim3(grid_for(bound, 8192)), dim3(BLOCK), 0, s, sc, b, in, out, hits, queue, counters,
                       shadow, res);
}

void launch_shade(hipStream_t s, int cls, uint32_t bound, const DevScene& sc, const BatchParams& b, const PathPool& in,
                  const PathPool& out, const HitRecords& hits, const uint32_t* queue, BounceCounters* counters,
                  const ShadowQueue& shadow, const SampleResults& res) {
    switch (cls) {
        case 0: launch_shade_t<0>(s, bound, sc, b, in, out, hits, queue, counters, shadow, res); break;
        case 1: launch_shade_t<1>(s, bound, sc, b, in, out, hits, queue, counters, shadow, res); break;
        case 2: launch_shade_t<2>(s, bound, sc, b, in, out, hits, queue, counters, shadow, res); break;
        case 3: launch_shade_t<3>(s, bound, sc, b, in, out, hits, queue, counters, shadow, res); break;
        case 4: launch_shade_t<4>(s, bound, sc, b, in, out, hits, queue, counters, shadow, res); break;
        case 5: launch_shade_t<5>(s, bound, sc, b, in, out, hits, queue, counters, shadow, res); break;
        case 6: launch_shade_t<6>(s, bound, sc, b, in, out, hits, queue, counters, shadow, res); break;
        case 7: launch_shade_t<7>(s, bound, sc, b, in, out, hits, queue, counters, shadow, res); break;
        default: break;
    }
}

void launch_connect(hipStream_t s, uint32_t bound, const DevScene& sc, const ShadowQueue& shadow, const BounceCounters* counters,
                    const SampleResults& res, unsigned long long* ray_counter) {
    hipLaunchKernelGGL(k_connect, dim3(grid_for(bound, 8192)), dim3(BLOCK), 0, s, sc, shadow, counters, res, ray_counter);
}

void launch_accumulate(hipStream_t s, const BatchParams& b, const SampleResults& res, double* image, double* position,
                       double* normal, unsigned long long* pixel_counters) {
    hipLaunchKernelGGL(k_accumulate, dim3(grid_for(b.n_pixels, 4096)), dim3(BLOCK), 0, s, b, res, image, position, normal,
                       pixel_counters);
}

void launch_gather_samples(hipStream_t s, const BatchParams& b, const SampleResults& res, size_t pixel_base, double* color,
                           double* position, double* normal) {
    hipLaunchKernelGGL(k_gather_samples, dim3(grid_for(b.n_paths, 4096)), dim3(BLOCK), 0, s, b, res, pixel_base, color,
                       position, normal);
}

void launch_primary(hipStream_t s, const DevScene& sc, const BatchParams& b, int32_t* object_index, int32_t* primitive_index,
                    double* distance, double* position, double* normal) {
    hipLaunchKernelGGL(k_primary, dim3(grid_for(b.n_pixels, 4096)), dim3(BLOCK), 0, s, sc, b, object_index, primitive_index,
                       distance, position, normal);
}

} // namespace ptk
