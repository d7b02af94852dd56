// Acceleration-structure build on the GPU: a linear BVH over Morton-ordered triangles, collapsed to the 4-wide nodes
// the traversal kernels read.
//
// Replaces, like the host builder (host/bvh_builder.cpp), BVH::Node::Node / surface_area_heuristic of the reference
// (src/bvh.cpp:91-165); any valid hierarchy yields the same closest hit, so only the build time and the cost of walking
// the tree differ.  Steps, each a kernel over triangles or nodes:
//   1. bounds     per triangle: its box, padded like the reference's (src/bvh.cpp:16,30) and rounded outward to fp32,
//                 and the bounds of all centroids (block reductions + ordered-integer atomics)
//   2. keys       63-bit Morton code of the centroid (21 bits per axis)
//   3. sort       radix sort of (key, triangle) pairs (rocPRIM)
//   4. topology   binary radix tree over the sorted keys, one thread per inner node (duplicate keys are told apart by
//                 their position, so the tree stays balanced where many centroids coincide)
//   5. fit        boxes bottom-up: the second child to arrive at a node (an atomic counter per node) computes it
//   6. collapse   top-down, one level per launch: a 4-wide node takes a binary node's two children and opens the ones
//                 with the largest surface area until it has four; subtrees of at most `max_leaf` triangles become
//                 leaves (their triangles are consecutive in the sorted order)
// The tree is worse than the host builder's surface-area tree (more nodes visited per ray); the build is two to three
// orders of magnitude faster.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <vector>

#include "../host/bvh_builder.h"
#include "scene_device.h"

namespace ptk {
namespace {

constexpr int BUILD_BLOCK = 256;

struct Box3f {
    float lo[3], hi[3];
};

__device__ __forceinline__ unsigned int ordered(float f) { // order-preserving map float -> uint
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float unordered(unsigned int u) {
    u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// 1. boxes of the triangles and the bounds of their centroids; centroid_bounds: [0..2] min, [3..5] max, as ordered uints
__global__ __launch_bounds__(BUILD_BLOCK) void k_build_bounds(const double* __restrict__ vertices, uint32_t n, Box3f* __restrict__ boxes,
                                                              float* __restrict__ centroids, unsigned int* centroid_bounds) {
    __shared__ float s_lo[3][BUILD_BLOCK / 64], s_hi[3][BUILD_BLOCK / 64];
    const uint32_t i = blockIdx.x * BUILD_BLOCK + threadIdx.x;
    float c[3] = {0, 0, 0};
    float clo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, chi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    if (i < n) {
        const double* p = vertices + 9 * static_cast<size_t>(i);
        Box3f b;
        for (int a = 0; a < 3; a++) {
            const double lo = fmin(p[a], fmin(p[3 + a], p[6 + a])), hi = fmax(p[a], fmax(p[3 + a], p[6 + a]));
            b.lo[a] = __double2float_rd(lo - EPSILON);
            b.hi[a] = __double2float_ru(hi + EPSILON);
            c[a] = static_cast<float>((p[a] + p[3 + a] + p[6 + a]) / 3.0);
            clo[a] = chi[a] = c[a];
        }
        boxes[i] = b;
        centroids[3 * static_cast<size_t>(i)] = c[0];
        centroids[3 * static_cast<size_t>(i) + 1] = c[1];
        centroids[3 * static_cast<size_t>(i) + 2] = c[2];
    }
    for (int a = 0; a < 3; a++) {
        for (int off = 32; off > 0; off >>= 1) {
            clo[a] = fminf(clo[a], __shfl_down(clo[a], off));
            chi[a] = fmaxf(chi[a], __shfl_down(chi[a], off));
        }
        if ((threadIdx.x & 63) == 0) {
            s_lo[a][threadIdx.x >> 6] = clo[a];
            s_hi[a][threadIdx.x >> 6] = chi[a];
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        float lo = s_lo[threadIdx.x][0], hi = s_hi[threadIdx.x][0];
        for (int w = 1; w < BUILD_BLOCK / 64; w++) {
            lo = fminf(lo, s_lo[threadIdx.x][w]);
            hi = fmaxf(hi, s_hi[threadIdx.x][w]);
        }
        atomicMin(centroid_bounds + threadIdx.x, ordered(lo));
        atomicMax(centroid_bounds + 3 + threadIdx.x, ordered(hi));
    }
}

__device__ __forceinline__ unsigned long long spread21(unsigned long long v) { // bit k of v -> bit 3k
    v &= 0x1FFFFFull;
    v = (v | (v << 32)) & 0x1F00000000FFFFull;
    v = (v | (v << 16)) & 0x1F0000FF0000FFull;
    v = (v | (v << 8)) & 0x100F00F00F00F00Full;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

// 2. Morton keys
__global__ __launch_bounds__(BUILD_BLOCK) void k_build_keys(const float* __restrict__ centroids, uint32_t n, const unsigned int* centroid_bounds,
                                                            unsigned long long* __restrict__ keys, uint32_t* __restrict__ index) {
    const uint32_t i = blockIdx.x * BUILD_BLOCK + threadIdx.x;
    if (i >= n) return;
    unsigned long long key = 0;
    for (int a = 0; a < 3; a++) {
        const float lo = unordered(centroid_bounds[a]), hi = unordered(centroid_bounds[3 + a]);
        const float extent = hi - lo;
        float t = extent > 0 ? (centroids[3 * static_cast<size_t>(i) + a] - lo) / extent : 0.0f;
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        const unsigned long long q = static_cast<unsigned long long>(fminf(t * 2097152.0f, 2097151.0f));
        key |= spread21(q) << a;
    }
    keys[i] = key;
    index[i] = i;
}

// Binary radix tree.  Node references: >= 0 inner node, < 0 leaf ~reference = position in the sorted order.
struct RadixTree {
    int32_t* left;        // [n - 1]
    int32_t* right;       // [n - 1]
    int32_t* parent;      // [n - 1], -1 for the root (node 0)
    int32_t* leaf_parent; // [n]
    uint32_t* first;      // [n - 1] first sorted position the node covers
    uint32_t* last;       // [n - 1] last one
    Box3f* box;           // [n - 1]
    uint32_t* arrived;    // [n - 1] children that have reported their boxes
};

// length of the common prefix of the keys at sorted positions i and j (equal keys: continued by the positions), -1 outside
__device__ __forceinline__ int common_prefix(const unsigned long long* keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = keys[i], b = keys[j];
    if (a != b) return __clzll(static_cast<long long>(a ^ b));
    return 64 + __clz(i ^ j);
}

// 4. topology (T. Karras, "Maximizing parallelism in the construction of BVHs, octrees, and k-d trees", 2012)
__global__ __launch_bounds__(BUILD_BLOCK) void k_build_topology(const unsigned long long* __restrict__ keys, int n, RadixTree t) {
    const int i = static_cast<int>(blockIdx.x * BUILD_BLOCK + threadIdx.x);
    if (i >= n - 1) return;
    const int d = common_prefix(keys, n, i, i + 1) > common_prefix(keys, n, i, i - 1) ? 1 : -1;
    const int d_min = common_prefix(keys, n, i, i - d);
    int l_max = 2;
    while (common_prefix(keys, n, i, i + l_max * d) > d_min) l_max *= 2;
    int l = 0;
    for (int s = l_max / 2; s >= 1; s /= 2)
        if (common_prefix(keys, n, i, i + (l + s) * d) > d_min) l += s;
    const int j = i + l * d;
    const int d_node = common_prefix(keys, n, i, j);
    int s = 0;
    int step = l;
    do {
        step = (step + 1) / 2;
        if (common_prefix(keys, n, i, i + (s + step) * d) > d_node) s += step;
    } while (step > 1);
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    const int32_t left = lo == gamma ? ~gamma : gamma;
    const int32_t right = hi == gamma + 1 ? ~(gamma + 1) : gamma + 1;
    t.left[i] = left;
    t.right[i] = right;
    t.first[i] = static_cast<uint32_t>(lo);
    t.last[i] = static_cast<uint32_t>(hi);
    if (left >= 0) t.parent[left] = i;
    else t.leaf_parent[~left] = i;
    if (right >= 0) t.parent[right] = i;
    else t.leaf_parent[~right] = i;
    if (i == 0) t.parent[0] = -1;
}

__device__ __forceinline__ Box3f box_union(const Box3f& a, const Box3f& b) {
    Box3f r;
    for (int k = 0; k < 3; k++) {
        r.lo[k] = fminf(a.lo[k], b.lo[k]);
        r.hi[k] = fmaxf(a.hi[k], b.hi[k]);
    }
    return r;
}

// 5. boxes, bottom-up from every leaf; sorted_boxes: triangle boxes in sorted order
__global__ __launch_bounds__(BUILD_BLOCK) void k_build_fit(const Box3f* __restrict__ sorted_boxes, int n, RadixTree t) {
    const int k = static_cast<int>(blockIdx.x * BUILD_BLOCK + threadIdx.x);
    if (k >= n) return;
    int node = t.leaf_parent[k];
    while (node >= 0) {
        __threadfence(); // the box this thread wrote for the child it came from is visible before it reports
        if (atomicAdd(t.arrived + node, 1u) == 0) return; // the first child to arrive leaves; the second goes on
        __threadfence();
        const int32_t l = t.left[node], r = t.right[node];
        const Box3f a = l >= 0 ? t.box[l] : sorted_boxes[~l];
        const Box3f b = r >= 0 ? t.box[r] : sorted_boxes[~r];
        t.box[node] = box_union(a, b);
        node = t.parent[node];
    }
}

__global__ __launch_bounds__(BUILD_BLOCK) void k_build_gather_boxes(const Box3f* __restrict__ boxes, const uint32_t* __restrict__ index, uint32_t n,
                                                                    Box3f* __restrict__ sorted_boxes) {
    const uint32_t i = blockIdx.x * BUILD_BLOCK + threadIdx.x;
    if (i < n) sorted_boxes[i] = boxes[index[i]];
}

struct CollapseItem {
    int32_t binary; // inner node of the radix tree
    int32_t wide;   // the 4-wide node that takes its place
};

__device__ __forceinline__ float half_area(const Box3f& b) {
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

// 6. one level of the collapse: every item of `in` becomes a 4-wide node; its inner children get nodes of their own
// (wide_count) and go to `out`.  counters: [0] items written to `out`, [1] 4-wide nodes allocated so far.
__global__ __launch_bounds__(BUILD_BLOCK) void k_build_collapse(const CollapseItem* __restrict__ in, uint32_t n_in, CollapseItem* __restrict__ out,
                                                                uint32_t* counters, RadixTree t, const Box3f* __restrict__ sorted_boxes,
                                                                int max_leaf, BvhNode4* __restrict__ wide) {
    const uint32_t i = blockIdx.x * BUILD_BLOCK + threadIdx.x;
    if (i >= n_in) return;
    const CollapseItem item = in[i];
    int32_t kids[BVH_WIDTH];
    int nk = 2;
    kids[0] = t.left[item.binary];
    kids[1] = t.right[item.binary];
    auto is_leaf = [&](int32_t ref) { return ref < 0 || static_cast<int>(t.last[ref] - t.first[ref]) + 1 <= max_leaf; };
    auto box_of = [&](int32_t ref) { return ref >= 0 ? t.box[ref] : sorted_boxes[~ref]; };
    while (nk < BVH_WIDTH) { // open the inner child with the largest surface area
        int pick = -1;
        float area = -1.0f;
        for (int k = 0; k < nk; k++) {
            if (is_leaf(kids[k])) continue;
            const float a = half_area(t.box[kids[k]]);
            if (a > area) {
                area = a;
                pick = k;
            }
        }
        if (pick < 0) break;
        const int32_t open = kids[pick];
        kids[pick] = t.left[open];
        kids[nk++] = t.right[open];
    }
    BvhNode4 node;
    for (int k = 0; k < BVH_WIDTH; k++) {
        node.child[k] = BVH_EMPTY;
        node.lo_x[k] = node.lo_y[k] = node.lo_z[k] = __builtin_inff();
        node.hi_x[k] = node.hi_y[k] = node.hi_z[k] = -__builtin_inff();
        node._pad[k] = 0;
    }
    for (int k = 0; k < nk; k++) {
        const int32_t ref = kids[k];
        const Box3f b = box_of(ref);
        node.lo_x[k] = b.lo[0];
        node.lo_y[k] = b.lo[1];
        node.lo_z[k] = b.lo[2];
        node.hi_x[k] = b.hi[0];
        node.hi_y[k] = b.hi[1];
        node.hi_z[k] = b.hi[2];
        if (ref < 0) node.child[k] = ~static_cast<int32_t>(static_cast<uint32_t>(~ref) * 8u); // one triangle
        else if (is_leaf(ref)) node.child[k] = ~static_cast<int32_t>(t.first[ref] * 8u + (t.last[ref] - t.first[ref]));
        else {
            const uint32_t slot = atomicAdd(counters + 1, 1u);
            node.child[k] = static_cast<int32_t>(slot);
            out[atomicAdd(counters, 1u)] = CollapseItem{ref, static_cast<int32_t>(slot)};
        }
    }
    wide[item.wide] = node;
}

// Entries the ordered walk's stack can hold inside the subtree of `node` (as Collapser::emit computes it for the host build).
int stack_need(const std::vector<BvhNode4>& nodes, int32_t node) {
    int children = 0, below = 0;
    for (int c = 0; c < BVH_WIDTH; c++) {
        const int32_t child = nodes[static_cast<size_t>(node)].child[c];
        if (child == BVH_EMPTY) continue;
        children++;
        if (child >= 0) below = std::max(below, stack_need(nodes, child));
    }
    return children - 1 + below;
}

#define BUILD_TRY(expr)                \
    do {                               \
        hipError_t e_ = (expr);        \
        if (e_ != hipSuccess) {        \
            error = e_;                \
            goto done;                 \
        }                              \
    } while (0)

} // namespace

// Builds the hierarchy of `n` triangles (9 doubles each, host memory) on the current device, on `stream`.  Returns false
// when the GPU build does not apply (fewer than 1024 triangles, a tree deeper than `stack_limit` allows, or a HIP
// error, then left in *error): the caller builds on the host instead.  build_ms: device time of steps 1 to 6.
// memory_limit: most device memory the build may allocate, 0 for no limit (a build that would need more ends with
// hipErrorOutOfMemory, like one the device has no room for).
bool build_bvh4_gpu(const double* vertices, int64_t n64, int max_leaf, int stack_limit, hipStream_t stream, BuiltBvh& result,
                    double* build_ms, hipError_t* error_out, size_t memory_limit) {
    hipError_t error = hipSuccess;
    bool ok = false;
    if (n64 < 1024 || n64 > (int64_t(1) << 28)) return false;
    const uint32_t n = static_cast<uint32_t>(n64);
    max_leaf = std::min(std::max(max_leaf, 1), 8);
    const uint32_t blocks = (n + BUILD_BLOCK - 1) / BUILD_BLOCK;
    std::vector<void*> allocations;
    size_t allocated = 0;
    auto alloc = [&](auto** p, size_t count) {
        void* q = nullptr;
        const size_t bytes = std::max<size_t>(count, 1) * sizeof(**p);
        allocated += bytes;
        // (memory_limit: the build's device memory is capped, as if the device had no more; 0 = no cap)
        const hipError_t e = memory_limit && allocated > memory_limit ? hipErrorOutOfMemory : hipMalloc(&q, bytes);
        if (e == hipSuccess) allocations.push_back(q);
        *p = static_cast<std::remove_reference_t<decltype(*p)>>(q);
        return e;
    };
    double* d_vertices = nullptr;
    Box3f *d_boxes = nullptr, *d_sorted_boxes = nullptr;
    float* d_centroids = nullptr;
    unsigned int* d_bounds = nullptr;
    unsigned long long *d_keys = nullptr, *d_keys_sorted = nullptr;
    uint32_t *d_index = nullptr, *d_index_sorted = nullptr, *d_counters = nullptr;
    void* d_temp = nullptr;
    size_t temp_bytes = 0;
    RadixTree tree{};
    CollapseItem* d_items[2] = {nullptr, nullptr};
    BvhNode4* d_wide = nullptr;
    hipEvent_t begin = nullptr, end = nullptr;
    uint32_t n_items = 1, n_wide = 1;
    int levels = 0;
    float ms = 0;
    const unsigned int init_bounds[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    const CollapseItem root{0, 0};
    uint32_t counters[2];

    BUILD_TRY(alloc(&d_vertices, static_cast<size_t>(n) * 9));
    BUILD_TRY(alloc(&d_boxes, n));
    BUILD_TRY(alloc(&d_sorted_boxes, n));
    BUILD_TRY(alloc(&d_centroids, static_cast<size_t>(n) * 3));
    BUILD_TRY(alloc(&d_bounds, 6));
    BUILD_TRY(alloc(&d_keys, n));
    BUILD_TRY(alloc(&d_keys_sorted, n));
    BUILD_TRY(alloc(&d_index, n));
    BUILD_TRY(alloc(&d_index_sorted, n));
    BUILD_TRY(alloc(&d_counters, 2));
    BUILD_TRY(alloc(&tree.left, n));
    BUILD_TRY(alloc(&tree.right, n));
    BUILD_TRY(alloc(&tree.parent, n));
    BUILD_TRY(alloc(&tree.leaf_parent, n));
    BUILD_TRY(alloc(&tree.first, n));
    BUILD_TRY(alloc(&tree.last, n));
    BUILD_TRY(alloc(&tree.box, n));
    BUILD_TRY(alloc(&tree.arrived, n));
    BUILD_TRY(alloc(&d_items[0], n));
    BUILD_TRY(alloc(&d_items[1], n));
    BUILD_TRY(alloc(&d_wide, n));
    BUILD_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, d_keys, d_keys_sorted, d_index, d_index_sorted, n, 0, 63, stream));
    {
        unsigned char* t = nullptr;
        BUILD_TRY(alloc(&t, temp_bytes));
        d_temp = t;
    }
    BUILD_TRY(hipMemcpyAsync(d_vertices, vertices, static_cast<size_t>(n) * 72, hipMemcpyHostToDevice, stream));
    BUILD_TRY(hipMemcpyAsync(d_bounds, init_bounds, sizeof(init_bounds), hipMemcpyHostToDevice, stream));
    BUILD_TRY(hipMemsetAsync(tree.arrived, 0, static_cast<size_t>(n) * sizeof(uint32_t), stream));
    BUILD_TRY(hipEventCreate(&begin));
    BUILD_TRY(hipEventCreate(&end));
    BUILD_TRY(hipEventRecord(begin, stream));
    hipLaunchKernelGGL(k_build_bounds, dim3(blocks), dim3(BUILD_BLOCK), 0, stream, d_vertices, n, d_boxes, d_centroids, d_bounds);
    hipLaunchKernelGGL(k_build_keys, dim3(blocks), dim3(BUILD_BLOCK), 0, stream, d_centroids, n, d_bounds, d_keys, d_index);
    BUILD_TRY(rocprim::radix_sort_pairs(d_temp, temp_bytes, d_keys, d_keys_sorted, d_index, d_index_sorted, n, 0, 63, stream));
    hipLaunchKernelGGL(k_build_gather_boxes, dim3(blocks), dim3(BUILD_BLOCK), 0, stream, d_boxes, d_index_sorted, n, d_sorted_boxes);
    hipLaunchKernelGGL(k_build_topology, dim3(blocks), dim3(BUILD_BLOCK), 0, stream, d_keys_sorted, static_cast<int>(n), tree);
    hipLaunchKernelGGL(k_build_fit, dim3(blocks), dim3(BUILD_BLOCK), 0, stream, d_sorted_boxes, static_cast<int>(n), tree);
    BUILD_TRY(hipGetLastError());
    // collapse, level by level: the host reads the size of the next level after every launch
    BUILD_TRY(hipMemcpyAsync(d_items[0], &root, sizeof(root), hipMemcpyHostToDevice, stream));
    while (n_items > 0) {
        counters[0] = 0;
        counters[1] = n_wide;
        BUILD_TRY(hipMemcpyAsync(d_counters, counters, sizeof(counters), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(k_build_collapse, dim3((n_items + BUILD_BLOCK - 1) / BUILD_BLOCK), dim3(BUILD_BLOCK), 0, stream, d_items[levels & 1],
                           n_items, d_items[(levels + 1) & 1], d_counters, tree, d_sorted_boxes, max_leaf, d_wide);
        BUILD_TRY(hipGetLastError());
        BUILD_TRY(hipMemcpyAsync(counters, d_counters, sizeof(counters), hipMemcpyDeviceToHost, stream));
        BUILD_TRY(hipStreamSynchronize(stream));
        n_items = counters[0];
        n_wide = counters[1];
        levels++;
        if (levels > 64) goto done; // deeper than any stack could follow
    }
    BUILD_TRY(hipEventRecord(end, stream));
    BUILD_TRY(hipEventSynchronize(end));
    BUILD_TRY(hipEventElapsedTime(&ms, begin, end));
    result.nodes.resize(n_wide);
    result.order.resize(n);
    BUILD_TRY(hipMemcpy(result.nodes.data(), d_wide, static_cast<size_t>(n_wide) * sizeof(BvhNode4), hipMemcpyDeviceToHost));
    BUILD_TRY(hipMemcpy(result.order.data(), d_index_sorted, static_cast<size_t>(n) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    result.max_depth = levels;
    result.max_stack = stack_need(result.nodes, 0);
    result.sah_depth_cap = 0;
    if (build_ms) *build_ms = ms;
    ok = result.max_stack <= stack_limit; // a tree the walk's stack cannot follow is built on the host instead
done:
    if (begin) (void) hipEventDestroy(begin);
    if (end) (void) hipEventDestroy(end);
    for (void* p : allocations) (void) hipFree(p);
    if (error_out) *error_out = error;
    return ok;
}

} // namespace ptk
