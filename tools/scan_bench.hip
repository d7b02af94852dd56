// The object scan of the shading kernels (scan_objects, geometry.h) on its own: a Cornell-like table of seven
// rectangles and five spheres, two rays per lane, at the occupancy the shading kernels run at (two waves per SIMD).
// Prints the time one pass over the table costs a SIMD, for the table as runs by type and as one general run.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Ipathtracing_amd/csrc/device
//              -Ipathtracing_amd/csrc tools/scan_bench.hip -o tools/scan_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "scene_device.h"
#include "geometry.h"

using namespace ptk;

#define CHECK(e)                                             \
    do {                                                     \
        hipError_t err_ = (e);                               \
        if (err_ != hipSuccess) {                            \
            printf("%s: %s\n", #e, hipGetErrorString(err_)); \
            return 1;                                        \
        }                                                    \
    } while (0)

constexpr int ROUNDS = 64;

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_scan(DevScene sc, double* out) {
    __shared__ double pad[8192]; // 64 KB: two blocks per compute unit, as for k_bounce
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    Rng rng{rng_key(7, id, 0), 0};
    double sum = pad[(threadIdx.x * 7) & 8191];
    TraversalStack st{};
    for (int r = 0; r < ROUNDS; r++) {
        const V3 org = v3(rng.uniform(-0.9, 0.9), rng.uniform(-0.9, 0.9), rng.uniform(-0.9, 0.9));
        const V3 d0 = normalize(v3(rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-1, 1)));
        const V3 d1 = normalize(v3(rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-1, 1)));
        RayState rays[2] = {ray_state<false>(sc, org, d0, INF), ray_state<false>(sc, org, d1, INF)};
#ifndef SKIP_SCAN
        scan_objects<false, MESH_INLINE, 2>(sc, org, rays, st);
#endif
        sum += rays[0].limit + rays[1].limit + rays[0].obj + rays[1].obj;
    }
    out[id] = sum;
}

struct Object {
    int kind;       // 0, 1, 2: wall with that normal axis; 3: sphere
    double p[3], r; // wall: point and half side; sphere: centre and radius
};
static Object sphere(double x, double y, double z, double r) { return Object{3, {x, y, z}, r}; }
static Object wall(int axis, double at, double half = 1.0) {
    Object o{axis, {0, 0, 0}, half};
    o.p[axis] = at;
    return o;
}
// the scan tables (ScanHeader) of the objects: in runs by type, or all of them as general shapes in table order
static std::vector<unsigned char> tables(const std::vector<Object>& objects, bool runs) {
    ScanHeader header{};
    std::vector<ShapeUnit> units[4];
    std::vector<DevShape> others;
    for (size_t i = 0; i < objects.size(); i++) {
        const Object& o = objects[i];
        if (runs) {
            ShapeUnit u{};
            u.index = static_cast<int32_t>(i);
            if (o.kind == 3) {
                memcpy(u.f, o.p, 24);
                u.f[3] = o.r;
            }
            else {
                u.f[0] = o.p[o.kind];
                u.f[1] = o.p[(o.kind + 1) % 3];
                u.f[2] = o.p[(o.kind + 2) % 3];
                u.f[3] = u.f[4] = o.r + EPSILON;
            }
            units[o.kind].push_back(u);
        }
        else {
            DevShape s{};
            s.index = static_cast<int32_t>(i);
            memcpy(s.f, o.p, 24);
            if (o.kind == 3) {
                s.type = PTK_OBJ_SPHERE;
                s.f[3] = o.r;
            }
            else {
                s.type = PTK_OBJ_RECTANGLE;
                s.f[3 + o.kind] = 1.0;
                s.f[6 + (o.kind + 1) % 3] = 1.0;
                s.f[9 + (o.kind + 2) % 3] = 1.0;
                s.f[12] = s.f[13] = o.r + EPSILON;
            }
            others.push_back(s);
        }
    }
    for (int k = 0; k < 3; k++) header.n_axis[k] = static_cast<int32_t>(units[k].size());
    header.n_spheres = static_cast<int32_t>(units[3].size());
    header.n_others = static_cast<int32_t>(others.size());
    std::vector<unsigned char> bytes(sizeof(header));
    memcpy(bytes.data(), &header, sizeof(header));
    for (int k = 0; k < 4; k++) bytes.insert(bytes.end(), (unsigned char*) units[k].data(), (unsigned char*) (units[k].data() + units[k].size()));
    bytes.insert(bytes.end(), (unsigned char*) others.data(), (unsigned char*) (others.data() + others.size()));
    return bytes;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 2 * 8;
    double* out;
    CHECK(hipMalloc(&out, sizeof(double) * 256 * blocks));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const std::vector<Object> objects = {wall(1, 0.999, 0.2), sphere(-0.5, -0.6, 0.3, 0.35), sphere(0.4, -0.6, -0.2, 0.35),
                                         sphere(0.0, -0.7, 0.5, 0.3), wall(1, 1.0), wall(2, -1.0), wall(0, -1.0),
                                         sphere(-0.3, 0.2, -0.4, 0.3), sphere(0.5, 0.3, 0.4, 0.25), wall(0, 1.0), wall(2, 1.0),
                                         wall(1, -1.0)};
    for (int form = 0; form < 2; form++) {
        const std::vector<unsigned char> bytes = tables(objects, form != 0);
        unsigned char* d_shapes;
        CHECK(hipMalloc(&d_shapes, bytes.size()));
        CHECK(hipMemcpy(d_shapes, bytes.data(), bytes.size(), hipMemcpyHostToDevice));
        DevScene sc{};
        sc.n_objects = static_cast<int32_t>(objects.size());
        sc.scan = reinterpret_cast<const ScanHeader*>(d_shapes);
        sc.has_axis_rectangles = form;
        float best = 1e30f;
        for (int rep = 0; rep < 5; rep++) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_scan, dim3(blocks), dim3(256), 0, 0, sc, out);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        // wave-rounds per SIMD = blocks * 4 waves * ROUNDS / (CUs * 4 SIMDs)
        const double per_simd = static_cast<double>(blocks) * ROUNDS / prop.multiProcessorCount;
        printf("%s: %.3f ms, %.0f ns of SIMD time per pass of one wave over the table (two rays, 12 objects)\n",
               form ? "runs by type" : "one general run", best, best * 1e6 / per_simd);
        CHECK(hipFree(d_shapes));
    }
    return 0;
}
