// Device inlines: ray/primitive tests, BVH traversal, surface attributes, textures.
// Reference code reproduced: src/objects.cpp (primitive tests, find_closest_hit, normals, UVs),
// src/bvh.cpp:60-83,179-237 (slab test, ordered traversal), src/valuemap.cpp:34-52.
#pragma once
#include "device_math.h"
#include "scene_device.h"

namespace ptk {

// Ray::prepare (include/utils.h:45-62): dominant axis and shear for the watertight triangle test.
struct RayPrep {
    int kx, ky, kz;
    double Sx, Sy, Sz;
};

PTK_DEV RayPrep prepare_ray(const V3& d) {
    RayPrep p;
    double ax = fabs(d.x), ay = fabs(d.y), az = fabs(d.z);
    int kz = 0;
    double m = ax;
    if (ay > m) {
        kz = 1;
        m = ay;
    }
    if (az > m) kz = 2;
    int kx = kz + 1;
    if (kx == 3) kx = 0;
    int ky = kx + 1;
    if (ky == 3) ky = 0;
    double dxp = comp(d, kx), dyp = comp(d, ky), dzp = comp(d, kz);
    p.kx = kx;
    p.ky = ky;
    p.kz = kz;
    p.Sx = -dxp / dzp;
    p.Sy = -dyp / dzp;
    p.Sz = 1.0 / dzp;
    return p;
}

PTK_DEV V3 permute(const V3& v, const RayPrep& p) {
    // kz selects a cyclic rotation of (x, y, z)
    if (p.kz == 2) return v;
    if (p.kz == 0) return V3{v.y, v.z, v.x};
    return V3{v.z, v.x, v.y};
}

// Triangle::find_closest_object_hit (src/objects.cpp:360-406).
PTK_DEV bool triangle_hit(const TriHot& t, const V3& o, const RayPrep& rp, double t_max, double& distance) {
    V3 a = permute(v3(t.p1[0] - o.x, t.p1[1] - o.y, t.p1[2] - o.z), rp);
    V3 b = permute(v3(t.p2[0] - o.x, t.p2[1] - o.y, t.p2[2] - o.z), rp);
    V3 c = permute(v3(t.p3[0] - o.x, t.p3[1] - o.y, t.p3[2] - o.z), rp);
    a.x += rp.Sx * a.z;
    a.y += rp.Sy * a.z;
    b.x += rp.Sx * b.z;
    b.y += rp.Sy * b.z;
    c.x += rp.Sx * c.z;
    c.y += rp.Sy * c.z;
    double e1 = b.x * c.y - b.y * c.x;
    double e2 = c.x * a.y - c.y * a.x;
    double e3 = a.x * b.y - a.y * b.x;
    if ((e1 < 0 || e2 < 0 || e3 < 0) && (e1 > 0 || e2 > 0 || e3 > 0)) return false;
    double det = e1 + e2 + e3;
    if (det == 0) return false;
    double t_scaled = e1 * (a.z * rp.Sz) + e2 * (b.z * rp.Sz) + e3 * (c.z * rp.Sz);
    if (det < 0 && (t_scaled >= 0 || t_scaled < t_max * det)) return false;
    if (det > 0 && (t_scaled <= 0 || t_scaled > t_max * det)) return false;
    distance = t_scaled / det;
    return true;
}

// Per-lane traversal stack in LDS, laid out [depth][thread] so that the lanes of a wave touch
// consecutive banks; entries beyond LDS_STACK_DEPTH spill to a private array.
struct StackEntry {
    int32_t node;
    float tnear;
};

// The LDS address space is part of the pointer type: through a generic pointer the pushes and pops
// would be flat instructions, which take the slower path of the vector memory unit.
typedef __attribute__((address_space(3))) StackEntry LdsStackEntry;
// The entries beyond the LDS part live in an array of their own (the caller's, see make_stack): were it a member,
// its run-time indexing would keep the whole struct, sp and the counters included, in scratch memory, and every
// push and pop would go through memory for them.
struct TraversalStack {
    LdsStackEntry* lds; // this thread's column: element d at lds[d * blockDim.x]
    int stride;
    StackEntry* spill;  // [SPILL_STACK_DEPTH]
    uint32_t* overflows; // device counter of pushes that found the stack full (uniform; stays zero, see push)
    int sp;
    uint32_t nodes_visited, triangles_tested; // statistics of k_traverse
    PTK_DEV void push(int32_t node, float tnear) {
        if (sp < LDS_STACK_DEPTH) {
            lds[sp * stride].node = node;
            lds[sp * stride].tnear = tnear;
        }
        else if (sp < LDS_STACK_DEPTH + SPILL_STACK_DEPTH) spill[sp - LDS_STACK_DEPTH] = StackEntry{node, tnear};
        else {
            // The host builds no tree that needs more entries than the stack has (build_bvh4 bounds the need and
            // ptk_scene_create checks it), so this is never reached; were it, the entry would be lost and the
            // count of such losses says so (PtkStats::stack_overflows).
            atomicAdd(overflows, 1u);
            return;
        }
        sp++;
    }
    PTK_DEV StackEntry pop() {
        sp--;
        if (sp < LDS_STACK_DEPTH) return StackEntry{lds[sp * stride].node, lds[sp * stride].tnear};
        return spill[sp - LDS_STACK_DEPTH];
    }
};

// Closest triangle of one mesh.  Children are visited nearest first and skipped when their entry
// distance is not below the best hit so far (Node::intersect, src/bvh.cpp:193-213).
//
// The slab test (BoundingBox::intersect, :60-83) only culls, so it runs in fp32 on the fp32 boxes with
// its rounding errors turned into a wider interval: t = (b - o) * (1/d) in fp32 is off by at most
// 2^-24 (|o| + |b|) |1/d| from rounding o and the difference, plus 3 * 2^-24 |t| from 1/d and the
// product.  Each axis interval is widened by 2^-22 (|o| + bound_abs) |1/d| and the final interval by
// a factor 1 -+ 2^-21; a box the exact test accepts is never rejected.  Triangles are tested exactly
// (fp64, the reference's watertight test).  The nearest child stays in a register; only the others
// go through the stack.
// bound_abs: largest absolute coordinate of the mesh's bounds.
// any_hit: return at the first triangle found below t_best (enough for a shadow ray against a mesh
// that no connection can pass through).
PTK_DEV void sort_pair(float& ta, int32_t& ca, float& tb, int32_t& cb) {
    const bool swap = tb < ta;
    const float t = swap ? tb : ta;
    const int32_t c = swap ? cb : ca;
    tb = swap ? ta : tb;
    cb = swap ? ca : cb;
    ta = t;
    ca = c;
}

typedef float Float2 __attribute__((ext_vector_type(2))); // two children at a time: v_pk_add_f32 / v_pk_mul_f32

// Next entry of the stack that can still hold something closer than t_best; BVH_EMPTY when there is none.
PTK_DEV int32_t pop_closer(TraversalStack& st, double t_best) {
    while (st.sp > 0) {
        const StackEntry e = st.pop();
        if (static_cast<double>(e.tnear) < t_best) return e.node;
    }
    return BVH_EMPTY;
}

// Per-ray constants of the fp32 slab test, two lanes wide (the same value in both halves).
struct SlabRay {
    Float2 nox, noy, noz; // -origin
    Float2 ix, iy, iz;    // 1 / direction
    Float2 px, py, pz;    // widening of each axis interval
};
PTK_DEV SlabRay slab_ray(const V3& org, const V3& dir, float bound_abs) {
    const float ox = static_cast<float>(org.x), oy = static_cast<float>(org.y), oz = static_cast<float>(org.z);
    const float idx = static_cast<float>(1.0 / dir.x), idy = static_cast<float>(1.0 / dir.y), idz = static_cast<float>(1.0 / dir.z);
    const float widen = 0x1p-22f;
    const float pad_x = widen * (fabsf(ox) + bound_abs) * fabsf(idx);
    const float pad_y = widen * (fabsf(oy) + bound_abs) * fabsf(idy);
    const float pad_z = widen * (fabsf(oz) + bound_abs) * fabsf(idz);
    SlabRay r;
    r.nox = Float2{-ox, -ox};
    r.noy = Float2{-oy, -oy};
    r.noz = Float2{-oz, -oz};
    r.ix = Float2{idx, idx};
    r.iy = Float2{idy, idy};
    r.iz = Float2{idz, idz};
    r.px = Float2{pad_x, pad_x};
    r.py = Float2{pad_y, pad_y};
    r.pz = Float2{pad_z, pad_z};
    return r;
}

// One inner node (already in registers: `n`): tests its four boxes, pushes the children that can hold something
// closer than t_best except the nearest, and returns the nearest; `pop` is set when no child qualifies (the caller
// then takes the next entry of the stack).
template<bool NARROW>
PTK_DEV int32_t visit_inner_children(const BvhNode4& n, const SlabRay& ray, double t_best, TraversalStack& st, bool& pop) {
    const float FAR = __builtin_inff();
    st.nodes_visited++;
    // NARROW (k_traverse): the ray's constants live as single registers between visits and are widened to pairs here, on
    // every visit; the empty asm keeps the compiler from widening them once, outside the walk's loop, where the pairs
    // would hold nine more registers for the whole walk (k_traverse then fits five waves per SIMD instead of four).
    SlabRay r = ray;
    if (NARROW) {
        float nox = ray.nox[0], noy = ray.noy[0], noz = ray.noz[0], ix = ray.ix[0], iy = ray.iy[0], iz = ray.iz[0], px = ray.px[0],
              py = ray.py[0], pz = ray.pz[0];
        asm volatile("" : "+v"(nox), "+v"(noy), "+v"(noz), "+v"(ix), "+v"(iy), "+v"(iz), "+v"(px), "+v"(py), "+v"(pz));
        r.nox = Float2{nox, nox};
        r.noy = Float2{noy, noy};
        r.noz = Float2{noz, noz};
        r.ix = Float2{ix, ix};
        r.iy = Float2{iy, iy};
        r.iz = Float2{iz, iz};
        r.px = Float2{px, px};
        r.py = Float2{py, py};
        r.pz = Float2{pz, pz};
    }
    float tn[BVH_WIDTH];
    int32_t ch[BVH_WIDTH];
#pragma unroll
    for (int h = 0; h < BVH_WIDTH / 2; h++) { // children 2h and 2h + 1
        const Float2 lx = {n.lo_x[2 * h], n.lo_x[2 * h + 1]}, hx = {n.hi_x[2 * h], n.hi_x[2 * h + 1]};
        const Float2 ly = {n.lo_y[2 * h], n.lo_y[2 * h + 1]}, hy = {n.hi_y[2 * h], n.hi_y[2 * h + 1]};
        const Float2 lz = {n.lo_z[2 * h], n.lo_z[2 * h + 1]}, hz = {n.hi_z[2 * h], n.hi_z[2 * h + 1]};
        const Float2 ax = (lx + r.nox) * r.ix, bx = (hx + r.nox) * r.ix;
        const Float2 ay = (ly + r.noy) * r.iy, by = (hy + r.noy) * r.iy;
        const Float2 az = (lz + r.noz) * r.iz, bz = (hz + r.noz) * r.iz;
        const Float2 near_x = __builtin_elementwise_min(ax, bx) - r.px, far_x = __builtin_elementwise_max(ax, bx) + r.px;
        const Float2 near_y = __builtin_elementwise_min(ay, by) - r.py, far_y = __builtin_elementwise_max(ay, by) + r.py;
        const Float2 near_z = __builtin_elementwise_min(az, bz) - r.pz, far_z = __builtin_elementwise_max(az, bz) + r.pz;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int c = 2 * h + k;
            float lo = fmaxf(fmaxf(0.0f, near_x[k]), fmaxf(near_y[k], near_z[k]));
            float hi = fminf(far_x[k], fminf(far_y[k], far_z[k]));
            lo *= 1.0f - 0x1p-21f;
            hi *= 1.0f + 0x1p-21f;
            ch[c] = n.child[c];
            const bool ok = ch[c] != BVH_EMPTY && hi >= lo && static_cast<double>(lo) < t_best;
            tn[c] = ok ? lo : FAR;
        }
    }
    if (BVH_WIDTH == 4) {
        sort_pair(tn[0], ch[0], tn[1], ch[1]);
        sort_pair(tn[2], ch[2], tn[3], ch[3]);
        sort_pair(tn[0], ch[0], tn[2], ch[2]);
        sort_pair(tn[1], ch[1], tn[3], ch[3]);
        sort_pair(tn[1], ch[1], tn[2], ch[2]);
    }
    else { // the 19-comparator network for eight
        constexpr int NET[19][2] = {{0, 1}, {2, 3}, {4, 5}, {6, 7}, {0, 2}, {1, 3}, {4, 6}, {5, 7}, {1, 2}, {5, 6},
                                    {0, 4}, {3, 7}, {1, 5}, {2, 6}, {1, 4}, {3, 6}, {2, 4}, {3, 5}, {3, 4}};
#pragma unroll
        for (int k = 0; k < 19; k++) sort_pair(tn[NET[k][0] % BVH_WIDTH], ch[NET[k][0] % BVH_WIDTH], tn[NET[k][1] % BVH_WIDTH], ch[NET[k][1] % BVH_WIDTH]);
    }
#pragma unroll
    for (int c = BVH_WIDTH - 1; c >= 1; c--)
        if (tn[c] < FAR) st.push(ch[c], tn[c]);
    pop = !(tn[0] < FAR);
    return ch[0];
}

// Every pass of the loop takes the ray one step further: the box tests of an inner node's four children, or one
// triangle of a leaf.  The lanes of a wave that hold an inner node and those that hold a leaf fetch with the same
// load instructions (a node is 112 bytes, a triangle 72: five 16-byte loads serve both, two more the nodes), so a
// pass waits for memory once.  (sc.tri_hot ends in one spare element: the fifth load reads 8 bytes past a triangle.)
// Returns node, tri_i for the next pass; node == BVH_EMPTY: the walk is over.
struct WalkStep {
    bool found;     // a triangle became the closest hit in this pass
    int32_t prim;   // that triangle
};
template<bool NARROW>
PTK_DEV WalkStep walk_step(const DevScene& sc, int32_t& node, int& tri_i, int64_t first_tri, const V3& org, const RayPrep& rp,
                           const SlabRay& slab, double& t_best, TraversalStack& st) {
    WalkStep out{false, -1};
    const bool is_inner = node >= 0;
    const uint32_t code = static_cast<uint32_t>(~node);
    const int64_t tri = first_tri + (code >> 3) + tri_i;
    const double* src = is_inner ? reinterpret_cast<const double*>(sc.nodes + node) : reinterpret_cast<const double*>(sc.tri_hot + tri);
    constexpr int NODE_LOADS = BVH_NODE_BYTES / 16;
    double raw[2 * NODE_LOADS];
#pragma unroll
    for (int k = 0; k < 5; k++) __builtin_memcpy(raw + 2 * k, src + 2 * k, 16);
    bool pop = false; // the lane's next node comes from its stack
    if (is_inner) {
#pragma unroll
        for (int k = 5; k < NODE_LOADS; k++) __builtin_memcpy(raw + 2 * k, src + 2 * k, 16);
        BvhNode4 nd;
        __builtin_memcpy(&nd, raw, BVH_NODE_BYTES);
        node = visit_inner_children<NARROW>(nd, slab, t_best, st, pop);
    }
    else {
        TriHot th;
        __builtin_memcpy(&th, raw, sizeof(TriHot));
        st.triangles_tested++;
        double d;
        if (triangle_hit(th, org, rp, t_best, d) && d > EPSILON && d < t_best) {
            t_best = d;
            out.found = true;
            out.prim = static_cast<int32_t>(tri);
        }
        if (++tri_i > static_cast<int>(code & 7u)) { // the leaf is done
            pop = true;
            tri_i = 0;
        }
    }
    if (pop) node = pop_closer(st, t_best);
    return out;
}

PTK_DEV void mesh_closest_hit(const DevScene& sc, int32_t bvh_root, int64_t first_tri, const V3& org, const V3& dir,
                              const RayPrep& rp, double& t_best, int& prim_best, TraversalStack& st, float bound_abs,
                              bool any_hit = false) {
    const SlabRay r = slab_ray(org, dir, bound_abs);
    st.sp = 0;
    int32_t node = bvh_root < 0 ? BVH_EMPTY : bvh_root; // (a mesh without triangles has no root)
    int tri_i = 0;
    while (node != BVH_EMPTY) {
        const WalkStep step = walk_step<false>(sc, node, tri_i, first_tri, org, rp, r, t_best, st);
        if (step.found) {
            prim_best = step.prim;
            if (any_hit) return;
        }
    }
}

// ---- four lanes per ray --------------------------------------------------------------------------------------
// k_traverse gives every ray a quad of lanes (lanes 4q .. 4q + 3 of the wave).  At an inner node lane c tests the box
// of child c; at a leaf lane c tests triangle c.  What a ray's lanes share (the ray, its closest hit, its node, its
// stack pointer) is kept identically in all four; values cross lanes with DPP quad permutes, which cost no memory
// access.  All control flow that contains a permute is uniform within a quad.
template<int CTRL>
PTK_DEV int quad_perm_i(int v) {
    return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true);
}
template<int CTRL>
PTK_DEV float quad_perm_f(float v) {
    return __int_as_float(quad_perm_i<CTRL>(__float_as_int(v)));
}
template<int CTRL>
PTK_DEV double quad_perm_d(double v) {
    return __hiloint2double(quad_perm_i<CTRL>(__double2hiint(v)), quad_perm_i<CTRL>(__double2loint(v)));
}
// lane c reads lane (c + k) % 4 of its quad
constexpr int QUAD_ROTATE_1 = 0x39, QUAD_ROTATE_2 = 0x4E, QUAD_ROTATE_3 = 0x93;
// sum over the quad (in every lane)
PTK_DEV int quad_sum(int v) {
    v += quad_perm_i<0xB1>(v); // neighbours swapped: [1, 0, 3, 2]
    v += quad_perm_i<QUAD_ROTATE_2>(v);
    return v;
}

// The ray's stack: one column of LDS per ray, entry d at stack[d * rays_per_block]; all four lanes hold sp.
struct QuadStack {
    LdsStackEntry* column;
    int stride;
    int sp;
    uint32_t* overflows;
};

// Inner node `node` for one ray: lane c tests child c; the children that can hold something closer than t_best are
// pushed farthest first except the nearest, which is returned (or the next entry of the stack that can still matter,
// BVH_EMPTY when there is none).  c: the lane's position in its quad.
PTK_DEV int32_t pop_closer_quad(QuadStack& st, double t_best) {
    while (st.sp > 0) {
        st.sp--;
        const int32_t node = st.column[st.sp * st.stride].node;
        const float tnear = st.column[st.sp * st.stride].tnear;
        if (static_cast<double>(tnear) < t_best) return node;
    }
    return BVH_EMPTY;
}

struct SlabRay1 { // per-ray constants of the fp32 slab test (see slab_ray)
    float nox, noy, noz, ix, iy, iz, px, py, pz;
};
PTK_DEV SlabRay1 slab_ray1(const V3& org, const V3& dir, float bound_abs) {
    const SlabRay r = slab_ray(org, dir, bound_abs);
    return SlabRay1{r.nox[0], r.noy[0], r.noz[0], r.ix[0], r.iy[0], r.iz[0], r.px[0], r.py[0], r.pz[0]};
}

PTK_DEV int32_t visit_inner_quad(const BvhNodeQuad* nodes, int32_t node, int c, const SlabRay1& r, double t_best, QuadStack& st) {
    const float FAR = __builtin_inff();
    const float4* box = reinterpret_cast<const float4*>(&nodes[node].c[c]);
    const float4 a = box[0], b = box[1]; // lo.x lo.y lo.z hi.x | hi.y hi.z child pad
    const int32_t child = __float_as_int(b.z);
    // the arithmetic of visit_inner, one box per lane
    const float ax = (a.x + r.nox) * r.ix, bx = (a.w + r.nox) * r.ix;
    const float ay = (a.y + r.noy) * r.iy, by = (b.x + r.noy) * r.iy;
    const float az = (a.z + r.noz) * r.iz, bz = (b.y + r.noz) * r.iz;
    float lo = fmaxf(fmaxf(0.0f, fminf(ax, bx) - r.px), fmaxf(fminf(ay, by) - r.py, fminf(az, bz) - r.pz));
    float hi = fminf(fmaxf(ax, bx) + r.px, fminf(fmaxf(ay, by) + r.py, fmaxf(az, bz) + r.pz));
    lo *= 1.0f - 0x1p-21f;
    hi *= 1.0f + 0x1p-21f;
    const bool ok = child != BVH_EMPTY && hi >= lo && static_cast<double>(lo) < t_best;
    const float t = ok ? lo : FAR;
    // rank of this child among the quad's hits: how many of the others come first (nearer, or as near in a lower lane)
    const float t1 = quad_perm_f<QUAD_ROTATE_1>(t), t2 = quad_perm_f<QUAD_ROTATE_2>(t), t3 = quad_perm_f<QUAD_ROTATE_3>(t);
    const int rank = ((t1 < t || (t1 == t && c + 1 >= 4)) ? 1 : 0) + ((t2 < t || (t2 == t && c + 2 >= 4)) ? 1 : 0) +
                     ((t3 < t || (t3 == t && c + 3 >= 4)) ? 1 : 0);
    const int n_hit = quad_sum(ok ? 1 : 0);
    if (n_hit == 0) return pop_closer_quad(st, t_best);
    // ranks 1 .. n_hit - 1 go to the stack, rank 1 on top
    if (ok && rank > 0) {
        const int at = st.sp + (n_hit - 1 - rank);
        if (at < QUAD_STACK_DEPTH) {
            st.column[at * st.stride].node = child;
            st.column[at * st.stride].tnear = t;
        }
        else atomicAdd(st.overflows, 1u); // never: the host bounds what a hierarchy needs (see TraversalStack::push)
    }
    st.sp = min(st.sp + n_hit - 1, QUAD_STACK_DEPTH);
    return quad_sum(ok && rank == 0 ? child : 0);
}

// Leaf `leaf` for one ray: lane c tests triangle c (and c + 4, ... for leaves of more than four).  Returns whether a
// triangle became the ray's closest hit; t_best and prim_best are updated in all four lanes.  Ties between the
// triangles of a leaf go to the first, as in the one-lane walk.
PTK_DEV bool visit_leaf_quad(const DevScene& sc, int32_t leaf, int64_t first_tri, int c, const V3& org, const RayPrep& rp, double& t_best,
                             int& prim_best) {
    const uint32_t code = static_cast<uint32_t>(~leaf);
    const int64_t first = first_tri + (code >> 3);
    const int count = static_cast<int>(code & 7u) + 1;
    bool found = false;
    for (int base = 0; base < count; base += 4) { // quad-uniform
        const int k = base + c;
        double d = INF;
        bool valid = false;
        if (k < count) {
            valid = triangle_hit(sc.tri_hot[first + k], org, rp, t_best, d) && d > EPSILON && d < t_best;
            if (!valid) d = INF;
        }
        const double d1 = quad_perm_d<QUAD_ROTATE_1>(d), d2 = quad_perm_d<QUAD_ROTATE_2>(d), d3 = quad_perm_d<QUAD_ROTATE_3>(d);
        const bool beaten = (d1 < d || (d1 == d && c + 1 >= 4)) || (d2 < d || (d2 == d && c + 2 >= 4)) || (d3 < d || (d3 == d && c + 3 >= 4));
        const bool winner = valid && !beaten;
        if (quad_sum(winner ? 1 : 0) != 0) {
            const int lo = quad_sum(winner ? __double2loint(d) : 0), hi = quad_sum(winner ? __double2hiint(d) : 0);
            t_best = __hiloint2double(hi, lo);
            prim_best = quad_sum(winner ? static_cast<int>(first + k) : 0);
            found = true;
        }
    }
    return found;
}

// largest absolute coordinate of a mesh's bounds (DevShape::f[0..5])
PTK_DEV float mesh_bound_abs(const DevShape& o) {
    double m = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) m = fmax(m, fabs(o.f[k]));
    return static_cast<float>(m) * (1.0f + 0x1p-20f);
}

struct HitInfo {
    double t;
    int obj;
    int prim; // global triangle index (leaf order) for triangles and meshes, 0 otherwise
    int cls;  // class of the object's material
};

// One object's DevShape through the constant address space: the index is wave-uniform, so the
// record arrives in SGPRs by scalar loads instead of costing vector loads and registers per lane.
PTK_DEV void load_unit(const PTK_CONSTANT ShapeUnit* p, ShapeUnit& s) {
#pragma unroll
    for (int k = 0; k < 7; k++) s.f[k] = p->f[k];
    s.index = p->index;
}
PTK_DEV void load_shape(const PTK_CONSTANT DevShape* p, DevShape& s) {
#pragma unroll
    for (int k = 0; k < 14; k++) s.f[k] = p->f[k];
    s.type = p->type;
    s.index = p->index;
    s.first_tri = p->first_tri;
    s.bvh_root = p->bvh_root;
}

// One object against one ray: the body of find_closest_hit's loop (src/objects.cpp:414-442) with the
// primitive tests inlined; updates the closest hit so far and the shrinking t_max (`limit`).  First
// object wins exact ties.  The analytic tests are written with selects instead of branches (same
// arithmetic, same results): with two rays per object the compiler can then interleave their
// dependent chains of fp64 operations.
struct RayState {
    V3 dir;
    double dir_dot_org;
    RayPrep rp;
    // A hit at distance d is taken when d < limit: the distance of the closest hit so far, and before any
    // hit the smallest double above the ray's t_max (the reference takes d <= t_max for the first hit and
    // d < closest afterwards).
    double limit;
    V3 inv;          // frcp of the direction's components, when the scene has axis-aligned rectangles (else unused)
    int obj, prim;   // closest hit so far: object (-1: none) and triangle (leaf order; 0 for other shapes)
    bool needs_mesh; // MESH_DEFER: the ray enters the bounds of some mesh before its limit
};
// What the object scan does with a mesh: walk its hierarchy, only test its bounds (and leave the
// walk to k_traverse), or nothing (the walk has been done).
enum MeshMode : int { MESH_INLINE = 0, MESH_DEFER = 1, MESH_SKIP = 2 };

// Entry distance of the ray into an axis-aligned box, +infinity when it misses (BoundingBox::intersect,
// src/bvh.cpp:60-83).
PTK_DEV double box_entry(const double* bounds, const V3& org, const V3& dir) {
    double t0 = (bounds[0] - org.x) / dir.x, t1 = (bounds[3] - org.x) / dir.x;
    double lo = fmax(0.0, fmin(t0, t1)), hi = fmax(t0, t1);
    t0 = (bounds[1] - org.y) / dir.y;
    t1 = (bounds[4] - org.y) / dir.y;
    lo = fmax(lo, fmin(t0, t1));
    hi = fmin(hi, fmax(t0, t1));
    t0 = (bounds[2] - org.z) / dir.z;
    t1 = (bounds[5] - org.z) / dir.z;
    lo = fmax(lo, fmin(t0, t1));
    hi = fmin(hi, fmax(t0, t1));
    return hi >= lo ? lo : INF;
}

PTK_DEV void accept_hit(RayState& r, bool success, double d, int index, int prim) {
    if (success && d > EPSILON && d < r.limit) {
        r.limit = d;
        r.obj = index;
        r.prim = prim;
    }
}

// smallest double above t (t itself when it is infinite)
PTK_DEV double next_above(double t) {
    if (!(fabs(t) < INF)) return t;
    long long bits = __double_as_longlong(t);
    if (t == 0) return 0x1p-1074;
    return __longlong_as_double(t > 0 ? bits + 1 : bits - 1);
}

// A rectangle whose normal is coordinate axis K and whose sides run along the other two (ShapeUnit of the run of
// axis K).  The general test (test_shape) gives the same bits for such a rectangle: its dot products with a unit axis
// pick one coordinate exactly, whatever the axis's sign, and fdiv(s n, -(d n)) == -fdiv(s, d) for n = +-1.  What is
// saved: the dot products, and the reciprocal of the division, which all rectangles of one axis share (RayState::inv).
template<int K, int N>
PTK_DEV void test_axis_rectangle(const ShapeUnit& o, const V3& org, RayState (&rays)[N]) {
    constexpr int J = (K + 1) % 3, M = (K + 2) % 3;
    const double sk = axis<K>(org) - o.f[0], sj = axis<J>(org) - o.f[1], sm = axis<M>(org) - o.f[2];
#pragma unroll
    for (int k = 0; k < N; k++) {
        const double dk = axis<K>(rays[k].dir);
        const double d = -fdiv_by(sk, dk, axis<K>(rays[k].inv));
        const bool outside = fabs(fma(axis<J>(rays[k].dir), d, sj)) > o.f[3] || fabs(fma(axis<M>(rays[k].dir), d, sm)) > o.f[4];
        accept_hit(rays[k], !(fabs(dk) < EPSILON) && !outside, d, o.index, 0);
    }
}

// Sphere::find_closest_object_hit (src/objects.cpp:106-119); f: centre and radius.
template<int N>
PTK_DEV void test_sphere(const double* f, int index, const V3& org, RayState (&rays)[N]) {
    V3 c = v3(f[0], f[1], f[2]);
    double cc = len2(c - org) - f[3] * f[3];
#pragma unroll
    for (int k = 0; k < N; k++) {
        double d;
        double b = 2 * (rays[k].dir_dot_org - dot(rays[k].dir, c));
        bool success = solve_quadratic_select(b, cc, d);
        accept_hit(rays[k], success, d, index, 0);
    }
}

// One run of ShapeUnits: KIND 0, 1, 2 rectangles along that axis, 3 spheres.  The record of unit i + 1 is requested
// before unit i is tested, so the scalar-load latency overlaps the arithmetic.
template<int KIND, int N>
PTK_DEV void scan_units(const PTK_CONSTANT ShapeUnit* units, int n, const V3& org, RayState (&rays)[N]) {
    if (n <= 0) return;
    ShapeUnit o;
    load_unit(units, o);
    // The first record must have arrived before the loop is entered: a load still pending at the loop header makes
    // the compiler wait at the top of every iteration, which also waits for the prefetch issued there.
    asm volatile("" : : "s"(o.index));
    for (int i = 0; i < n; i++) {
        ShapeUnit next;
        load_unit(units + (i + 1 < n ? i + 1 : i), next);
        if (KIND == 3) test_sphere<N>(o.f, o.index, org, rays);
        else test_axis_rectangle<KIND, N>(o, org, rays);
        o = next;
    }
}

template<bool TRIS, int MODE, int N>
PTK_DEV void test_shape(const DevScene& sc, const DevShape& o, const V3& org, RayState (&rays)[N], TraversalStack& st) {
    const int index = o.index;
    if (o.type == PTK_OBJ_SPHERE) test_sphere<N>(o.f, index, org, rays);
    else if (o.type == PTK_OBJ_PLANE || o.type == PTK_OBJ_RECTANGLE) { // :185-211, :237-255
        V3 nrm = v3(o.f[3], o.f[4], o.f[5]);
        V3 shifted = org - v3(o.f[0], o.f[1], o.f[2]);
        double shifted_dot_n = dot(shifted, nrm);
        double d[N];
        bool success[N];
#pragma unroll
        for (int k = 0; k < N; k++) {
            double ddn = -dot(rays[k].dir, nrm);
            d[k] = fdiv(shifted_dot_n, ddn); // a zero or tiny ddn is rejected below whatever the quotient
            success[k] = !(fabs(ddn) < EPSILON); // d >= EPSILON and d <= t_max are implied by accept_hit's tests
        }
        if (o.type == PTK_OBJ_RECTANGLE) {
            V3 a1 = v3(o.f[6], o.f[7], o.f[8]), a2 = v3(o.f[9], o.f[10], o.f[11]);
            double s1 = dot(shifted, a1), s2 = dot(shifted, a2);
            double half1 = o.f[12], half2 = o.f[13]; // L / 2 + EPSILON
#pragma unroll
            for (int k = 0; k < N; k++) {
                bool outside = fabs(fma(dot(rays[k].dir, a1), d[k], s1)) > half1 || fabs(fma(dot(rays[k].dir, a2), d[k], s2)) > half2;
                success[k] = success[k] && !outside;
            }
        }
#pragma unroll
        for (int k = 0; k < N; k++) accept_hit(rays[k], success[k], d[k], index, 0);
    }
    else if (TRIS && o.type == PTK_OBJ_TRIANGLE) {
#pragma unroll
        for (int k = 0; k < N; k++) {
            double d = INF;
            bool success = triangle_hit(sc.tri_hot[o.first_tri], org, rays[k].rp, rays[k].limit, d);
            accept_hit(rays[k], success, d, index, o.first_tri);
        }
    }
    else if (TRIS && MODE == MESH_DEFER) { // mesh: bounds only
#pragma unroll
        for (int k = 0; k < N; k++)
            if (o.bvh_root >= 0 && box_entry(o.f, org, rays[k].dir) < rays[k].limit) rays[k].needs_mesh = true;
    }
    else if (TRIS && MODE == MESH_INLINE) { // mesh
#pragma unroll
        for (int k = 0; k < N; k++) {
            double tb = rays[k].limit;
            int pb = -1;
            mesh_closest_hit(sc, o.bvh_root, o.first_tri, org, rays[k].dir, rays[k].rp, tb, pb, st, mesh_bound_abs(o));
            accept_hit(rays[k], pb >= 0, tb, index, pb);
        }
    }
}

PTK_DEV void no_hit(HitInfo& hit) {
    hit.t = INF;
    hit.obj = -1;
    hit.prim = 0;
    hit.cls = 0;
}

// find_closest_hit (src/objects.cpp:414-442) without the attribute computation: the linear scan
// over top-level objects, for N rays that leave one origin together (a shadow ray and an extension
// ray leave every shaded vertex): one fetch of every object record, shared per-object terms, and N
// independent chains of arithmetic per object.  A ray that is not wanted is given
// t_max = -infinity, which no primitive test accepts.  The record of object i + 1 is requested
// before object i is tested, so the scalar-load latency overlaps the arithmetic.
// TRIS = false compiles the triangle test and the BVH traversal (and with it the stack) out, for
// scenes made of spheres, planes and rectangles only.
template<bool TRIS, int MODE, int N>
PTK_DEV void scan_objects(const DevScene& sc, const V3& org, RayState (&rays)[N], TraversalStack& st) {
    const PTK_CONSTANT ScanHeader* header = (const PTK_CONSTANT ScanHeader*) sc.scan;
    const PTK_CONSTANT ShapeUnit* units = (const PTK_CONSTANT ShapeUnit*) (header + 1);
    const int n_x = header->n_axis[0], n_y = header->n_axis[1], n_z = header->n_axis[2], n_spheres = header->n_spheres;
    scan_units<0, N>(units, n_x, org, rays);
    units += n_x;
    scan_units<1, N>(units, n_y, org, rays);
    units += n_y;
    scan_units<2, N>(units, n_z, org, rays);
    units += n_z;
    scan_units<3, N>(units, n_spheres, org, rays);
    const PTK_CONSTANT DevShape* shapes = (const PTK_CONSTANT DevShape*) (units + n_spheres);
    const int n = header->n_others;
    if (n <= 0) return;
    DevShape o;
    load_shape(shapes, o);
    asm volatile("" : : "s"(o.type)); // (as in scan_units)
    for (int i = 0; i < n; i++) {
        DevShape next;
        load_shape(shapes + (i + 1 < n ? i + 1 : i), next);
        test_shape<TRIS, MODE, N>(sc, o, org, rays, st);
        o = next;
    }
}

template<bool TRIS>
PTK_DEV RayState ray_state(const DevScene& sc, const V3& org, const V3& dir, double t_max) {
    RayState r;
    r.dir = dir;
    r.inv = v3(0.0);
    if (sc.has_axis_rectangles) r.inv = v3(frcp(dir.x), frcp(dir.y), frcp(dir.z));
    r.dir_dot_org = dot(dir, org);
    r.rp = RayPrep{};
    if (TRIS) r.rp = prepare_ray(dir);
    r.limit = next_above(t_max);
    r.obj = -1;
    r.prim = 0;
    r.needs_mesh = false;
    return r;
}

// The closest hit a finished scan left in the ray's state.
PTK_DEV HitInfo hit_of(const DevScene& sc, const RayState& r) {
    HitInfo h;
    no_hit(h);
    if (r.obj >= 0) {
        h.t = r.limit;
        h.obj = r.obj;
        h.prim = r.prim;
        h.cls = sc.objects[r.obj].mat.type; // the class of a surface vertex is its material type
    }
    return h;
}

template<bool TRIS>
PTK_DEV bool closest_hit(const DevScene& sc, const V3& org, const V3& dir, double t_max, HitInfo& hit, TraversalStack& st) {
    RayState rays[1] = {ray_state<TRIS>(sc, org, dir, t_max)};
    scan_objects<TRIS, MESH_INLINE, 1>(sc, org, rays, st);
    hit = hit_of(sc, rays[0]);
    return hit.obj >= 0;
}

// ---- surface attributes --------------------------------------------------------------------
PTK_DEV V3 tri_barycentric(const TriCold& t, const V3& p) { // compute_barycentric :346-353
    double x = dot(p, v3(t.v1)), y = dot(p, v3(t.v2));
    double l1 = ((t.y2 - t.y3) * (x - t.x3) + (t.x3 - t.x2) * (y - t.y3)) / t.det_T;
    double l2 = ((t.y3 - t.y1) * (x - t.x3) + (t.x1 - t.x3) * (y - t.y3)) / t.det_T;
    return v3(l1, l2, 1.0 - l1 - l2);
}

template<bool TRIS>
PTK_DEV V3 object_normal(const DevScene& sc, const DevObject& o, const V3& p, int prim) { // get_normal_vector
    if (o.type == PTK_OBJ_SPHERE) return normalize(p - v3(o.px, o.py, o.pz));
    if (!TRIS || o.type == PTK_OBJ_PLANE || o.type == PTK_OBJ_RECTANGLE) return v3(o.nx, o.ny, o.nz);
    const TriCold& t = sc.tri_cold[prim];
    if (t.smooth) { // :334-337
        V3 b = tri_barycentric(t, p);
        return normalize(v3(t.n1) * b.x + v3(t.n2) * b.y + v3(t.n3) * b.z);
    }
    return v3(t.normal);
}

template<bool TRIS>
PTK_DEV void object_uv(const DevScene& sc, const DevObject& o, const V3& p, int prim, double& u, double& v) { // get_UV
    if (o.type == PTK_OBJ_SPHERE) { // :96-104
        V3 unit = (p - v3(o.px, o.py, o.pz)) / o.radius;
        u = 0.5 + atan2(-unit.z, -unit.x) / (2 * PI);
        v = 0.5 + asin(-unit.y) / PI;
    }
    else if (o.type == PTK_OBJ_PLANE) { // :178-183
        V3 s = p - v3(o.px, o.py, o.pz);
        u = 1 - dot(s, v3(o.v1x, o.v1y, o.v1z)) - 0.5;
        v = 1 - dot(s, v3(o.v2x, o.v2y, o.v2z)) - 0.5;
    }
    else if (!TRIS || o.type == PTK_OBJ_RECTANGLE) { // :230-235
        V3 s = p - v3(o.px, o.py, o.pz);
        u = 1 - dot(s, v3(o.v1x, o.v1y, o.v1z)) / o.L1 - 0.5;
        v = 1 - dot(s, v3(o.v2x, o.v2y, o.v2z)) / o.L2 - 0.5;
    }
    else { // :355-358
        const TriCold& t = sc.tri_cold[prim];
        V3 b = tri_barycentric(t, p);
        u = t.uv1[0] * b.x + t.uv2[0] * b.y + t.uv3[0] * b.z;
        v = t.uv1[1] * b.x + t.uv2[1] * b.y + t.uv3[1] * b.z;
    }
}

// ---- textures (src/valuemap.cpp:34-52) -----------------------------------------------------
// Nearest texel, u wrapped, v flipped and wrapped.  1x1 maps skip the address arithmetic (and
// therefore the reference's "NaN coordinate reads as 0" rule, which needs a NaN UV to matter).
PTK_DEV int64_t texel_index(const DevMap& m, double u, double v) {
    int64_t ui = static_cast<int64_t>(static_cast<double>(m.width) * pos_fmod1(u));
    int64_t vi = static_cast<int64_t>(static_cast<double>(m.height) * pos_fmod1(1 - v));
    ui = ui < m.width ? ui : m.width - 1; // the products can round up to the size itself
    vi = vi < m.height ? vi : m.height - 1;
    return vi * m.width + ui;
}
PTK_DEV double tex1(const DevScene& sc, int id, bool use_uv, double u, double v) {
    if (id < 0) return 0.0;
    const DevMap& m = sc.maps[id];
    if (!use_uv || (m.width == 1 && m.height == 1)) return sc.map_data[m.offset];
    if (isnan(u) || isnan(v)) return 0.0;
    return sc.map_data[m.offset + texel_index(m, u, v)];
}
PTK_DEV V3 tex3(const DevScene& sc, int id, bool use_uv, double u, double v) {
    if (id < 0) return v3(0.0);
    const DevMap& m = sc.maps[id];
    int64_t i = 0;
    if (use_uv && !(m.width == 1 && m.height == 1)) {
        if (isnan(u) || isnan(v)) return v3(0.0);
        i = 3 * texel_index(m, u, v);
    }
    const double* p = sc.map_data + m.offset + i;
    return v3(p[0], p[1], p[2]);
}

} // namespace ptk
