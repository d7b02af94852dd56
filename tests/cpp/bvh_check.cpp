// Checks the host BVH builder's output on random triangles: every child reference is in range, every triangle is
// in exactly one leaf, every node is reachable and child boxes contain their subtrees.  usage: bvh_check <n_triangles> <max_leaf> <seed>; prints "ok ..." or a complaint.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "host/bvh_builder.h"

using namespace ptk;

struct Box3 {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
};

static long long visited_leaves = 0;
static std::vector<int> seen;

static bool walk(const BuiltBvh& b, const double* v, int32_t node, int depth, Box3& box, std::vector<int>& depth_of) {
    if (node < 0 || node >= static_cast<int32_t>(b.nodes.size())) {
        std::printf("node %d out of range\n", node);
        return false;
    }
    depth_of[static_cast<size_t>(node)] = depth;
    const BvhNode4& n = b.nodes[static_cast<size_t>(node)];
    for (int c = 0; c < 4; c++) {
        if (n.child[c] == BVH_EMPTY) continue;
        Box3 sub;
        if (n.child[c] >= 0) {
            if (!walk(b, v, n.child[c], depth + 1, sub, depth_of)) return false;
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
    return true;
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? std::atoll(argv[1]) : 1000;
    const int max_leaf = argc > 2 ? std::atoi(argv[2]) : 2;
    const unsigned seed = argc > 3 ? static_cast<unsigned>(std::atoi(argv[3])) : 1u;
    std::mt19937 gen(seed);
    std::uniform_real_distribution<double> pos(-1.0, 1.0), size(-0.05, 0.05);
    std::vector<double> v(static_cast<size_t>(n) * 9);
    for (int64_t i = 0; i < n; i++) {
        const double c[3] = {pos(gen), pos(gen), pos(gen)};
        for (int j = 0; j < 3; j++)
            for (int a = 0; a < 3; a++) v[static_cast<size_t>(9 * i + 3 * j + a)] = c[a] + size(gen);
    }
    BuiltBvh b = build_bvh4(v.data(), n, max_leaf, 8, 4);
    if (n == 0) {
        std::printf(b.nodes.empty() ? "ok empty\n" : "nodes for an empty mesh\n");
        return b.nodes.empty() ? 0 : 1;
    }
    seen.assign(static_cast<size_t>(n), 0);
    std::vector<int> depth_of(b.nodes.size(), -1);
    Box3 box;
    if (!walk(b, v.data(), 0, 1, box, depth_of)) return 1;
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
    std::printf("ok nodes=%zu depth=%d leaves=%lld\n", b.nodes.size(), b.max_depth, visited_leaves);
    return 0;
}
