// Checks the host BVH builder's output on random triangles: every child reference is in range, every triangle is
// in exactly one leaf, every node is reachable, child boxes contain their subtrees, and the stack the ordered walk
// needs (recomputed here from the tree's shape) is what the builder reports and within the limit asked for.
// usage: bvh_check <n_triangles> <max_leaf> <seed> [uniform|clustered|sliver] [stack_limit]; prints "ok ..." or a complaint.
// clustered: tight clusters of tiny triangles plus long thin ones that span the scene; sliver: triangles of
// geometrically growing size nested around one point (the surface-area heuristic peels them off one at a time).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "host/bvh_builder.h"

using namespace ptk;

struct Box3 {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
};

static long long visited_leaves = 0;
static std::vector<int> seen;

static bool walk(const BuiltBvh& b, const double* v, int32_t node, int depth, Box3& box, std::vector<int>& depth_of, int& stack_need) {
    if (node < 0 || node >= static_cast<int32_t>(b.nodes.size())) {
        std::printf("node %d out of range\n", node);
        return false;
    }
    depth_of[static_cast<size_t>(node)] = depth;
    const BvhNode4& n = b.nodes[static_cast<size_t>(node)];
    int children = 0, below = 0;
    for (int c = 0; c < BVH_WIDTH; c++) {
        if (n.child[c] == BVH_EMPTY) continue;
        children++;
        Box3 sub;
        if (n.child[c] >= 0) {
            int need = 0;
            if (!walk(b, v, n.child[c], depth + 1, sub, depth_of, need)) return false;
            below = std::max(below, need);
        }
        else {
            const uint32_t code = static_cast<uint32_t>(~n.child[c]);
            const uint32_t first = code >> 3, count = (code & 7u) + 1;
            visited_leaves++;
            for (uint32_t k = first; k < first + count; k++) {
                if (k >= b.order.size()) {
                    std::printf("leaf range out of bounds\n");
                    return false;
                }
                seen[b.order[k]]++;
                const double* p = v + 9 * static_cast<size_t>(b.order[k]);
                for (int j = 0; j < 3; j++)
                    for (int a = 0; a < 3; a++) {
                        sub.lo[a] = std::min(sub.lo[a], p[3 * j + a]);
                        sub.hi[a] = std::max(sub.hi[a], p[3 * j + a]);
                    }
            }
        }
        const float lo[3] = {n.lo_x[c], n.lo_y[c], n.lo_z[c]}, hi[3] = {n.hi_x[c], n.hi_y[c], n.hi_z[c]};
        for (int a = 0; a < 3; a++) {
            if (static_cast<double>(lo[a]) > sub.lo[a] || static_cast<double>(hi[a]) < sub.hi[a]) {
                std::printf("box of child %d of node %d does not contain its subtree\n", c, node);
                return false;
            }
            box.lo[a] = std::min(box.lo[a], sub.lo[a]);
            box.hi[a] = std::max(box.hi[a], sub.hi[a]);
        }
    }
    stack_need = children - 1 + below; // every child but the one the walk descends into waits on the stack
    return true;
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? std::atoll(argv[1]) : 1000;
    const int max_leaf = argc > 2 ? std::atoi(argv[2]) : 2;
    const unsigned seed = argc > 3 ? static_cast<unsigned>(std::atoi(argv[3])) : 1u;
    const char* shape = argc > 4 ? argv[4] : "uniform";
    const int stack_limit = argc > 5 ? std::atoi(argv[5]) : 64;
    std::mt19937 gen(seed);
    std::uniform_real_distribution<double> pos(-1.0, 1.0), size(-0.05, 0.05);
    std::vector<double> v(static_cast<size_t>(n) * 9);
    for (int64_t i = 0; i < n; i++) {
        double c[3] = {pos(gen), pos(gen), pos(gen)};
        double extent = 1.0;
        if (std::strcmp(shape, "clustered") == 0) {
            if (i % 16 == 0) extent = 40.0; // a long thin triangle through the scene
            else {
                const int cluster = static_cast<int>(i % 7);
                for (int a = 0; a < 3; a++) c[a] = 0.9 * std::sin(cluster * (a + 1.3)) + 1e-3 * pos(gen);
                extent = 1e-4;
            }
        }
        else if (std::strcmp(shape, "sliver") == 0) {
            for (int a = 0; a < 3; a++) c[a] = 0;
            extent = std::pow(1.35, static_cast<double>(i % 120)) * 1e-9;
        }
        for (int j = 0; j < 3; j++)
            for (int a = 0; a < 3; a++) {
                double d = size(gen) * extent;
                if (extent > 1.0 && a != static_cast<int>(i / 16 % 3)) d *= 1e-4; // thin: long along one axis only
                v[static_cast<size_t>(9 * i + 3 * j + a)] = c[a] + d;
            }
    }
#ifndef PTK_BVH_CHECK_THREADS
#define PTK_BVH_CHECK_THREADS 4
#endif
    BuiltBvh b = build_bvh4(v.data(), n, max_leaf, 8, PTK_BVH_CHECK_THREADS, stack_limit);
    if (n == 0) {
        std::printf(b.nodes.empty() ? "ok empty\n" : "nodes for an empty mesh\n");
        return b.nodes.empty() ? 0 : 1;
    }
    seen.assign(static_cast<size_t>(n), 0);
    std::vector<int> depth_of(b.nodes.size(), -1);
    Box3 box;
    int stack_need = 0;
    if (!walk(b, v.data(), 0, 1, box, depth_of, stack_need)) return 1;
    if (n > 1 || b.max_stack != 0) {
        if (stack_need != b.max_stack) {
            std::printf("the builder reports a stack need of %d, the tree's shape gives %d\n", b.max_stack, stack_need);
            return 1;
        }
    }
    if (b.max_stack > stack_limit && b.sah_depth_cap != 0) {
        std::printf("stack need %d above the limit %d although a shallower build was possible\n", b.max_stack, stack_limit);
        return 1;
    }
    if (b.max_stack > 64) {
        std::printf("stack need %d above what the kernels hold (64)\n", b.max_stack);
        return 1;
    }
    for (int64_t i = 0; i < n; i++)
        if (seen[static_cast<size_t>(i)] != 1) {
            std::printf("triangle %lld is in %d leaves\n", static_cast<long long>(i), seen[static_cast<size_t>(i)]);
            return 1;
        }
    for (size_t i = 0; i < b.nodes.size(); i++)
        if (depth_of[i] < 0) {
            std::printf("node %zu is unreachable\n", i);
            return 1;
        }
    std::printf("ok nodes=%zu depth=%d stack=%d sah_levels=%d leaves=%lld\n", b.nodes.size(), b.max_depth, b.max_stack, b.sah_depth_cap,
                visited_leaves);
    return 0;
}
