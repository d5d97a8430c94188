// Closest hit through the LBVH, with the tie rule of the reference's linear scan.
#pragma once
#include "rrt_device.hpp"

namespace rrt {

// The reference's scan (src/scene.rs:35-47, src/shapes/mesh.rs:28-89) visits primitives in a fixed
// order and lets sphere / plane / quad hits replace an earlier hit at EQUAL t (their range test is
// t > t_max -> reject) while a triangle does not (t < t_max).  A tree visits primitives in another
// order, so the winner among primitives at exactly the closest t is reconstructed from two numbers:
//   first = the lowest scan index at that t, last_incl = the highest scan index of an inclusive type
// -> the scan would end on last_incl if it comes after first, else on first.
template <typename R>
struct HitAcc {
    R t;
    int first;
    int last_incl;
};

template <typename R>
RRT_DEV void hit_init(HitAcc<R>& h) {
    h.t = Num<R>::tmax();
    h.first = -1;
    h.last_incl = -1;
}

template <typename R>
RRT_DEV void hit_consider(HitAcc<R>& h, R t, int g, bool inclusive) {
    if (t < h.t) {
        h.t = t;
        h.first = g;
        h.last_incl = inclusive ? g : -1;
    } else if (t == h.t) {
        h.first = g < h.first ? g : h.first;
        if (inclusive) h.last_incl = g > h.last_incl ? g : h.last_incl;
    }
}

template <typename R>
RRT_DEV int hit_winner(const HitAcc<R>& h) {
    return h.last_incl > h.first ? h.last_incl : h.first;
}

// Ray/box slab test for the two children of a node.  Conservative: boxes are padded on the host and
// the exit distance is widened by a few ulps, so the only effect of rounding is an extra visit.
template <typename R>
RRT_DEV void slab2(const BvhNode& N, V3<R> o, V3<R> inv_d, R t_best, bool& hit0, bool& hit1, R& near0, R& near1) {
    const R widen = sizeof(R) == 4 ? R(1.0000004) : R(1.0000000001);
    R ax = ((R)N.lo0[0] - o.x) * inv_d.x, bx = ((R)N.hi0[0] - o.x) * inv_d.x;
    R ay = ((R)N.lo0[1] - o.y) * inv_d.y, by = ((R)N.hi0[1] - o.y) * inv_d.y;
    R az = ((R)N.lo0[2] - o.z) * inv_d.z, bz = ((R)N.hi0[2] - o.z) * inv_d.z;
    R en = fmax(fmax(fmin(ax, bx), fmin(ay, by)), fmin(az, bz));
    R ex = fmin(fmin(fmax(ax, bx), fmax(ay, by)), fmax(az, bz)) * widen;
    hit0 = en <= ex && ex >= R(0) && en <= t_best;
    near0 = en;
    ax = ((R)N.lo1[0] - o.x) * inv_d.x; bx = ((R)N.hi1[0] - o.x) * inv_d.x;
    ay = ((R)N.lo1[1] - o.y) * inv_d.y; by = ((R)N.hi1[1] - o.y) * inv_d.y;
    az = ((R)N.lo1[2] - o.z) * inv_d.z; bz = ((R)N.hi1[2] - o.z) * inv_d.z;
    en = fmax(fmax(fmin(ax, bx), fmin(ay, by)), fmin(az, bz));
    ex = fmin(fmin(fmax(ax, bx), fmax(ay, by)), fmax(az, bz)) * widen;
    hit1 = en <= ex && ex >= R(0) && en <= t_best;
    near1 = en;
}

constexpr int BVH_STACK = 64;  // a radix tree over 64-bit keys is at most 64 levels deep

// Scene::intersect through the tree (plus a scan of the primitives kept out of it).
template <typename R>
RRT_DEV int closest_hit_bvh(const SceneView<R>& sv, V3<R> o, V3<R> d, R& t_out, HitCounters& hc) {
    HitAcc<R> acc;
    hit_init(acc);
    const PrimRec<R> RRT_CONSTANT* prims = as_constant(sv.prims);
    // unbounded primitives (planes): wave-uniform scan
    const int32_t RRT_CONSTANT* lin = as_constant(sv.linear_prim);
    for (int k = 0; k < sv.n_linear_prims; ++k) {
        const int g = lin[k];
        R t;
        if (prim_test(prims + g, o, d, Num<R>::tmax(), t)) hit_consider(acc, t, g, prims[g].type != PRIM_TRIANGLE);
        hc.prim_tests += 1;
    }
    if (sv.n_bvh_prims == 1) {
        const int g = sv.bvh_prim[0];
        R t;
        if (prim_test(prims + g, o, d, Num<R>::tmax(), t)) hit_consider(acc, t, g, prims[g].type != PRIM_TRIANGLE);
        hc.prim_tests += 1;
    } else if (sv.n_bvh_prims > 1) {
        // 1/d with the sign of d kept for zero components: +-inf slabs work with min/max that drop NaN
        V3<R> inv_d = mk<R>(R(1) / d.x, R(1) / d.y, R(1) / d.z);
        int stack[BVH_STACK];
        int sp = 0;
        int node = 0;
        for (;;) {
            if (node >= 0) {
                const BvhNode N = sv.nodes[node];
                bool h0, h1;
                R n0, n1;
                slab2<R>(N, o, inv_d, acc.t, h0, h1, n0, n1);
                hc.node_tests += 1;
                if (h0 && h1) {
                    const bool first0 = n0 <= n1;
                    if (sp < BVH_STACK) stack[sp++] = first0 ? N.c1 : N.c0;
                    node = first0 ? N.c0 : N.c1;
                    continue;
                }
                if (h0 || h1) {
                    node = h0 ? N.c0 : N.c1;
                    continue;
                }
            } else {
                const int g = sv.bvh_prim[~node];
                R t;
                if (prim_test(prims + g, o, d, Num<R>::tmax(), t)) hit_consider(acc, t, g, prims[g].type != PRIM_TRIANGLE);
                hc.prim_tests += 1;
            }
            if (sp == 0) break;
            node = stack[--sp];
        }
    }
    t_out = acc.t;
    return hit_winner(acc);
}

// Closest hit over the whole scene.  Returns the scan-order primitive index or -1.
template <typename R, bool BVH>
RRT_DEV int closest_hit(const SceneView<R>& sv, V3<R> o, V3<R> d, R& t, HitCounters& hc) {
    if (BVH) return closest_hit_bvh<R>(sv, o, d, t, hc);
    R best_t = Num<R>::tmax();
    int best_g = -1;
    scan_linear(sv, o, d, best_t, best_g, hc);
    t = best_t;
    return best_g;
}

}  // namespace rrt
