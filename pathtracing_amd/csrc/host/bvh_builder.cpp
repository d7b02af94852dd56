#include "bvh_builder.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <future>
#include <limits>

namespace ptk {
namespace {

struct Box {
    double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX};
    double hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    void grow(const Box& b) {
        for (int a = 0; a < 3; a++) {
            lo[a] = std::min(lo[a], b.lo[a]);
            hi[a] = std::max(hi[a], b.hi[a]);
        }
    }
    void grow(const double* p) {
        for (int a = 0; a < 3; a++) {
            lo[a] = std::min(lo[a], p[a]);
            hi[a] = std::max(hi[a], p[a]);
        }
    }
    double half_area() const {
        double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Node2 {
    Box box;
    int32_t left = -1, right = -1; // inner
    uint32_t first = 0, count = 0; // leaf: range in the index array
};

struct Builder {
    const std::vector<Box>& boxes;
    const std::vector<double>& centroids; // [n][3]
    std::vector<uint32_t>& index;
    std::vector<Node2>& nodes;
    std::atomic<int32_t>& next_node;
    int max_leaf, n_bins;
    int sah_depth_cap; // below this depth splits follow the surface-area heuristic, from it on they halve the range

    int32_t build(uint32_t first, uint32_t count, int depth, int parallel_levels) {
        const int32_t id = next_node.fetch_add(1);
        Box box, cbox;
        for (uint32_t i = first; i < first + count; i++) {
            box.grow(boxes[index[i]]);
            cbox.grow(&centroids[3 * static_cast<size_t>(index[i])]);
        }
        nodes[static_cast<size_t>(id)].box = box;
        if (count <= static_cast<uint32_t>(max_leaf)) {
            nodes[static_cast<size_t>(id)].first = first;
            nodes[static_cast<size_t>(id)].count = count;
            return id;
        }
        // the axis of the median split (and of the surface-area search when only one axis is searched): the longest
        int axis = 0;
        double extent = cbox.hi[0] - cbox.lo[0];
        for (int a = 1; a < 3; a++) {
            if (cbox.hi[a] - cbox.lo[a] > extent) {
                axis = a;
                extent = cbox.hi[a] - cbox.lo[a];
            }
        }
        uint32_t mid = first + count / 2;
        bool split_done = false;
        if (extent > 0 && depth < sah_depth_cap) {
            // Binned surface-area heuristic over all three axes in one pass over the triangles: the cheapest of the
            // 3 * (n_bins - 1) candidate planes (cost = area * count of either side).
            const int B = n_bins;
            std::vector<Box> bin_box(static_cast<size_t>(3 * B));
            std::vector<uint32_t> bin_count(static_cast<size_t>(3 * B), 0);
            double scale[3];
            for (int a = 0; a < 3; a++) scale[a] = cbox.hi[a] > cbox.lo[a] ? B / (cbox.hi[a] - cbox.lo[a]) : 0.0;
            auto bin_of = [&](uint32_t tri, int a) {
                int b = static_cast<int>((centroids[3 * static_cast<size_t>(tri) + a] - cbox.lo[a]) * scale[a]);
                return std::min(std::max(b, 0), B - 1);
            };
            for (uint32_t i = first; i < first + count; i++) {
                const uint32_t tri = index[i];
                for (int a = 0; a < 3; a++) {
                    const size_t slot = static_cast<size_t>(a * B + bin_of(tri, a));
                    bin_box[slot].grow(boxes[tri]);
                    bin_count[slot]++;
                }
            }
            double best = std::numeric_limits<double>::infinity();
            int best_bin = -1, best_axis = axis;
            std::vector<double> right_area(static_cast<size_t>(B), 0.0);
            for (int a = 0; a < 3; a++) {
                if (scale[a] == 0.0) continue;
                Box acc;
                for (int b = B - 1; b > 0; b--) {
                    if (bin_count[static_cast<size_t>(a * B + b)]) acc.grow(bin_box[static_cast<size_t>(a * B + b)]);
                    right_area[static_cast<size_t>(b)] = acc.half_area();
                }
                Box lacc;
                uint32_t lcount = 0;
                for (int b = 0; b < B - 1; b++) {
                    if (bin_count[static_cast<size_t>(a * B + b)]) lacc.grow(bin_box[static_cast<size_t>(a * B + b)]);
                    lcount += bin_count[static_cast<size_t>(a * B + b)];
                    const uint32_t rcount = count - lcount;
                    if (lcount == 0 || rcount == 0) continue;
                    const double cost = lacc.half_area() * lcount + right_area[static_cast<size_t>(b) + 1] * rcount;
                    if (cost < best) {
                        best = cost;
                        best_bin = b;
                        best_axis = a;
                    }
                }
            }
            if (best_bin >= 0) {
                auto it = std::partition(index.begin() + first, index.begin() + first + count,
                                         [&](uint32_t tri) { return bin_of(tri, best_axis) <= best_bin; });
                mid = static_cast<uint32_t>(it - index.begin());
                split_done = mid > first && mid < first + count;
            }
        }
        if (!split_done) { // degenerate centroid distribution or depth guard: split at the median
            mid = first + count / 2;
            std::nth_element(index.begin() + first, index.begin() + mid, index.begin() + first + count,
                             [&](uint32_t a, uint32_t b) {
                                 return centroids[3 * static_cast<size_t>(a) + axis] < centroids[3 * static_cast<size_t>(b) + axis];
                             });
        }
        int32_t l, r;
        if (parallel_levels > 0 && count > 65536) {
            auto fut = std::async(std::launch::async, [&] { return build(first, mid - first, depth + 1, parallel_levels - 1); });
            r = build(mid, first + count - mid, depth + 1, parallel_levels - 1);
            l = fut.get();
        }
        else {
            l = build(first, mid - first, depth + 1, 0);
            r = build(mid, first + count - mid, depth + 1, 0);
        }
        nodes[static_cast<size_t>(id)].left = l;
        nodes[static_cast<size_t>(id)].right = r;
        return id;
    }
};

float round_down(double v) {
    float f = static_cast<float>(v);
    if (static_cast<double>(f) > v) f = std::nextafter(f, -std::numeric_limits<float>::infinity());
    return f;
}
float round_up(double v) {
    float f = static_cast<float>(v);
    if (static_cast<double>(f) < v) f = std::nextafter(f, std::numeric_limits<float>::infinity());
    return f;
}

struct Collapser {
    const std::vector<Node2>& n2;
    std::vector<BvhNode4>& out;
    bool level_order; // open the children of a node level by level instead of by surface area
    int max_depth = 0;

    int32_t encode_leaf(const Node2& n) const {
        return ~static_cast<int32_t>(n.first * 8u + (n.count - 1u));
    }

    // Emits the 4-wide node for binary node `id` and its subtree.  stack_need: the most entries the ordered walk
    // (visit_inner in device/geometry.h) can hold while it is inside this subtree: a node pushes every child but
    // the one it descends into.
    int32_t emit(int32_t id, int depth, int& stack_need) {
        max_depth = std::max(max_depth, depth);
        const int32_t me = static_cast<int32_t>(out.size());
        out.emplace_back();
        int32_t kids[BVH_WIDTH];
        int level[BVH_WIDTH]; // binary levels below `id`
        int nk = 0;
        kids[nk] = n2[static_cast<size_t>(id)].left;
        level[nk++] = 1;
        kids[nk] = n2[static_cast<size_t>(id)].right;
        level[nk++] = 1;
        while (nk < BVH_WIDTH) { // open the inner child with the largest surface area (level_order: the shallowest first)
            int pick = -1;
            double area = -1;
            for (int i = 0; i < nk; i++) {
                const Node2& c = n2[static_cast<size_t>(kids[i])];
                if (c.left < 0) continue;
                const bool better = pick < 0 || (level_order && level[i] < level[pick]) ||
                                    ((!level_order || level[i] == level[pick]) && c.box.half_area() > area);
                if (better) {
                    area = c.box.half_area();
                    pick = i;
                }
            }
            if (pick < 0) break;
            const Node2& c = n2[static_cast<size_t>(kids[pick])];
            kids[pick] = c.left;
            level[pick]++;
            kids[nk] = c.right;
            level[nk++] = level[pick];
        }
        BvhNode4 node{};
        for (int i = 0; i < BVH_WIDTH; i++) {
            node.child[i] = BVH_EMPTY;
            node.lo_x[i] = node.lo_y[i] = node.lo_z[i] = std::numeric_limits<float>::infinity();
            node.hi_x[i] = node.hi_y[i] = node.hi_z[i] = -std::numeric_limits<float>::infinity();
        }
        for (int i = 0; i < nk; i++) {
            const Node2& c = n2[static_cast<size_t>(kids[i])];
            // boxes padded like the reference's (src/bvh.cpp:16,30) and rounded outward to fp32
            node.lo_x[i] = round_down(c.box.lo[0] - EPSILON);
            node.lo_y[i] = round_down(c.box.lo[1] - EPSILON);
            node.lo_z[i] = round_down(c.box.lo[2] - EPSILON);
            node.hi_x[i] = round_up(c.box.hi[0] + EPSILON);
            node.hi_y[i] = round_up(c.box.hi[1] + EPSILON);
            node.hi_z[i] = round_up(c.box.hi[2] + EPSILON);
        }
        out[static_cast<size_t>(me)] = node;
        int below = 0;
        for (int i = 0; i < nk; i++) {
            const Node2& c = n2[static_cast<size_t>(kids[i])];
            int need = 0;
            int32_t ref = c.left >= 0 ? emit(kids[i], depth + 1, need) : encode_leaf(c);
            below = std::max(below, need);
            out[static_cast<size_t>(me)].child[i] = ref;
        }
        stack_need = nk - 1 + below;
        return me;
    }
};

} // namespace

// Binary tree -> 4-wide nodes in result.nodes, with the tree's depth and stack need.
static void build_wide(const std::vector<Node2>& nodes, int32_t root, bool level_order, BuiltBvh& result) {
    result.nodes.clear();
    if (nodes[static_cast<size_t>(root)].left < 0) { // the whole mesh fits one leaf: root with a single leaf child
        BvhNode4 node{};
        const Node2& c = nodes[static_cast<size_t>(root)];
        for (int i = 0; i < BVH_WIDTH; i++) {
            node.child[i] = BVH_EMPTY;
            node.lo_x[i] = node.lo_y[i] = node.lo_z[i] = std::numeric_limits<float>::infinity();
            node.hi_x[i] = node.hi_y[i] = node.hi_z[i] = -std::numeric_limits<float>::infinity();
        }
        node.lo_x[0] = round_down(c.box.lo[0] - EPSILON);
        node.lo_y[0] = round_down(c.box.lo[1] - EPSILON);
        node.lo_z[0] = round_down(c.box.lo[2] - EPSILON);
        node.hi_x[0] = round_up(c.box.hi[0] + EPSILON);
        node.hi_y[0] = round_up(c.box.hi[1] + EPSILON);
        node.hi_z[0] = round_up(c.box.hi[2] + EPSILON);
        node.child[0] = ~static_cast<int32_t>(c.first * 8u + (c.count - 1u));
        result.nodes.push_back(node);
        result.max_depth = 1;
        result.max_stack = 0;
        return;
    }
    result.nodes.reserve(nodes.size() / 2);
    Collapser collapser{nodes, result.nodes, level_order};
    int need = 0;
    collapser.emit(root, 1, need);
    result.max_depth = collapser.max_depth;
    result.max_stack = need;
}

BuiltBvh build_bvh4(const double* vertices, int64_t n, int max_leaf, int n_bins, int threads, int stack_limit) {
    BuiltBvh result;
    if (n <= 0) return result;
    max_leaf = std::min(std::max(max_leaf, 1), 8);
    n_bins = std::min(std::max(n_bins, 2), 64);
    std::vector<Box> boxes(static_cast<size_t>(n));
    std::vector<double> centroids(static_cast<size_t>(n) * 3);
    for (int64_t i = 0; i < n; i++) {
        const double* p = vertices + 9 * i;
        Box b;
        b.grow(p);
        b.grow(p + 3);
        b.grow(p + 6);
        boxes[static_cast<size_t>(i)] = b;
        for (int a = 0; a < 3; a++) centroids[3 * static_cast<size_t>(i) + a] = (p[a] + p[3 + a] + p[6 + a]) / 3.0;
    }
    result.order.resize(static_cast<size_t>(n));
    std::vector<Node2> nodes(static_cast<size_t>(2 * n));
    int levels = 0;
    while ((1 << levels) < threads) levels++;
    // The walk's stack is finite, so the tree must not need more of it than `stack_limit` entries.  Triangles spread
    // evenly (every mesh seen so far) stay far below it; for a mesh the heuristic turns into a deep, lopsided tree the
    // build is repeated with the heuristic confined to fewer top levels, and at last with none: splitting every range
    // in halves and opening children level by level gives a 4-wide tree of depth ceil(log2(n) / 2) <= 14 that
    // needs at most 42 entries.
    const int caps[] = {56, 24, 12, 0};
    for (int attempt = 0; attempt < 4; attempt++) {
        for (int64_t i = 0; i < n; i++) result.order[static_cast<size_t>(i)] = static_cast<uint32_t>(i);
        std::fill(nodes.begin(), nodes.end(), Node2{});
        std::atomic<int32_t> next_node{0};
        Builder builder{boxes, centroids, result.order, nodes, next_node, max_leaf, n_bins, caps[attempt]};
        const int32_t root = builder.build(0, static_cast<uint32_t>(n), 0, levels);
        result.sah_depth_cap = caps[attempt];
        build_wide(nodes, root, caps[attempt] == 0, result);
        if (result.max_stack <= stack_limit) break;
    }
    return result;
}

} // namespace ptk
