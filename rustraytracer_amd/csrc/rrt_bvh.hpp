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

// (tracked queries: two accepted hits whose distances differ by less than their error bounds are an undecided order)
template <typename R, bool TRACK = false>
RRT_DEV void hit_consider(HitAcc<R>& h, R t, int g, bool inclusive, HitTrack<TRACK>* tr = nullptr) {
    if constexpr (TRACK) {
        if (h.first >= 0 && !(__builtin_fabsf(t - h.t) > tr->last_err + tr->best_err)) tr->amb = true;
        if (t < h.t) tr->best_err = tr->last_err;
    }
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

// Traversal stack of node links: the first BVH_LDS_LEVELS entries of every lane live in LDS
// ([level][thread], so the 64 lanes of a wave hit 64 different banks), deeper entries in a private
// spill array that a well-formed tree rarely reaches (a divergent private-memory access costs as
// much as a divergent global one).  A 4-wide node pushes at most three entries per level it descends; neither
// builder bounds the depth of its tree by construction, so the host measures it after every build and admits only
// trees of at most BVH_STACK_ENTRIES / 3 levels (rrt_types.hpp).  A push onto a full stack -- which such a tree
// cannot cause -- drops the link and counts it (SceneView::stack_overflow) rather than writing out of bounds.
// Entries carry no distance: a stale entry costs one node fetch whose children
// all fail, which happens for ~5 % of the nodes, and half the bytes per entry keep twice as many
// levels in LDS at full occupancy.
constexpr int BVH_BLOCK = 256;  // every kernel that traverses the tree runs 256-thread blocks
constexpr int BVH_DONE = BVH_NO_CHILD;

struct TreeStackLds {
    int node[BVH_LDS_LEVELS][BVH_BLOCK];
};
// The first BVH_TOP_NODES wide nodes of the array -- the top of the tree, both builders number top-down -- are kept in
// LDS by every kernel that walks the tree: a ray's first few node steps (all of them in a tree of a few hundred
// primitives) then read LDS instead of going through the vector memory path.  7.5 KB per block; the node stack keeps 12
// levels in LDS beside it (16 without it made no difference).  tree_top_load() is called by all threads of a block once,
// before the first walk.
constexpr int BVH_TOP_NODES = 120;
struct TreeTopLds {
    uint4 q[BVH_TOP_NODES][4];
};
static __shared__ TreeTopLds g_tree_top;

template <typename R>
RRT_DEV int tree_top_count(const SceneView<R>& sv) {
    return sv.n_bvh_prims - 1 < BVH_TOP_NODES ? sv.n_bvh_prims - 1 : BVH_TOP_NODES;  // (<= 0: no nodes)
}

template <typename R>
RRT_DEV void tree_top_load(const SceneView<R>& sv) {
    const int top_n = tree_top_count(sv);
    for (int t = threadIdx.x; t < top_n * 4; t += BVH_BLOCK) g_tree_top.q[t >> 2][t & 3] = reinterpret_cast<const uint4*>(sv.nodes)[t];
    __syncthreads();
}

// The four 16-byte words of wide node `node`.
template <typename R>
RRT_DEV void tree_node_words(const SceneView<R>& sv, int node, int top_n, uint4& q0, uint4& q1, uint4& q2, uint4& q3) {
    if (node < top_n) {
        q0 = g_tree_top.q[node][0]; q1 = g_tree_top.q[node][1]; q2 = g_tree_top.q[node][2]; q3 = g_tree_top.q[node][3];
    } else {
        const uint4* rec = reinterpret_cast<const uint4*>(sv.nodes + node);
        q0 = rec[0]; q1 = rec[1]; q2 = rec[2]; q3 = rec[3];
    }
}
// The private part of the stack is a separate local array in every kernel (a struct with a
// dynamically indexed member would live in private memory as a whole).
typedef int TreeStackSpill[BVH_SPILL_LEVELS];

// Per-lane traversal state.  node >= 0: a wide node to fetch; BVH_DONE: nothing left;
// any other negative value: a leaf to test.
template <typename R>
struct TreeWalk {
    V3<R> o, d, inv_d;
    HitAcc<R> acc;
    int node;
    int sp;
};

template <typename R>
RRT_DEV void walk_start(TreeWalk<R>& w, V3<R> o, V3<R> d) {
    w.o = o;
    w.d = d;
    // 1/d for the slab tests, with |d| floored at a value far below any direction cosine that matters:
    // a ray whose component is exactly zero is treated as tilted by 1e-18 (a shift of 1e-18 t, nothing
    // beside the padding of the boxes), and every product in the slab arithmetic stays finite -- an
    // inf or NaN there would switch the culling of that axis off, and one ray that visits a whole
    // million-node tree takes longer than a million ordinary rays.
    const R tiny = sizeof(R) == 4 ? R(1e-18) : R(1e-150);
    const R dx = Num<R>::abs(d.x) < tiny ? (d.x < R(0) ? -tiny : tiny) : d.x;
    const R dy = Num<R>::abs(d.y) < tiny ? (d.y < R(0) ? -tiny : tiny) : d.y;
    const R dz = Num<R>::abs(d.z) < tiny ? (d.z < R(0) ? -tiny : tiny) : d.z;
    w.inv_d = mk<R>(R(1) / dx, R(1) / dy, R(1) / dz);
    w.sp = 0;
    w.node = 0;
}

template <typename R>
RRT_DEV void walk_push(TreeWalk<R>& w, TreeStackLds& lds, TreeStackSpill& spill, int node, unsigned long long* overflow) {
    if (w.sp < BVH_LDS_LEVELS) {
        lds.node[w.sp][threadIdx.x] = node;
    } else if (w.sp < BVH_STACK_ENTRIES) {
        spill[w.sp - BVH_LDS_LEVELS] = node;
    } else {
        atomicAdd(overflow, 1ull);
        return;
    }
    w.sp += 1;
}

template <typename R>
RRT_DEV void walk_pop(TreeWalk<R>& w, TreeStackLds& lds, TreeStackSpill& spill) {
    if (w.sp == 0) {
        w.node = BVH_DONE;
        return;
    }
    w.sp -= 1;
    w.node = w.sp < BVH_LDS_LEVELS ? lds.node[w.sp][threadIdx.x] : spill[w.sp - BVH_LDS_LEVELS];
}

template <typename R>
RRT_DEV void order2(R& da, int& ca, R& db, int& cb) {
    const bool swap = db < da;
    const R d0 = swap ? db : da, d1 = swap ? da : db;
    const int c0 = swap ? cb : ca, c1 = swap ? ca : cb;
    da = d0; db = d1; ca = c0; cb = c1;
}

RRT_DEV float byte_to_float(uint32_t word, int k) {  // v_cvt_f32_ubyte<k>
    return (float)((word >> (8 * k)) & 0xffu);
}

// One wide node, given as the four 16-byte words of its WideNode record: slab tests of its (up to)
// four children.  dist[k] = entry distance of child k, or +inf if the ray misses it (or the slot is empty).
// Conservative: boxes are padded on the host, quantised outward by the builder, and the exit
// distance is widened by a few ulps, so the only effect of rounding is an extra visit.
// t_limit: boxes that begin beyond it are skipped (the closest distance so far; a tracked query adds that distance's
// error bound, so that no box is skipped for a distance that f64 might find larger)
template <typename R>
RRT_DEV void node_children(const TreeWalk<R>& w, uint4 q0, uint4 q1, uint4 q2, uint4 q3, R (&dist)[4], int (&child)[4], R t_limit) {
    const R widen = sizeof(R) == 4 ? R(1.0000004) : R(1.0000000001);
    const R inf = Num<R>::tmax();
    // plane distance t = (org + q s - o) / d = q (s / d) + (org - o) / d: two multiplies and a subtract per
    // axis and node, then one convert and one FMA per plane.  Rounding differs from decoding the plane
    // first by a few ulps of the ray-origin-to-node distance; the builder's padding is sized for that.
    const R sx = (R)__uint_as_float((q0.w & 0xffu) << 23), sy = (R)__uint_as_float(((q0.w >> 8) & 0xffu) << 23),
            sz = (R)__uint_as_float(((q0.w >> 16) & 0xffu) << 23);
    const R ax_ = sx * w.inv_d.x, ay_ = sy * w.inv_d.y, az_ = sz * w.inv_d.z;
    const R bx_ = ((R)__uint_as_float(q0.x) - w.o.x) * w.inv_d.x;
    const R by_ = ((R)__uint_as_float(q0.y) - w.o.y) * w.inv_d.y;
    const R bz_ = ((R)__uint_as_float(q0.z) - w.o.z) * w.inv_d.z;
    // Which bound of a slab the ray meets first depends on the sign of its direction only, so the entry ("near") and
    // exit ("far") planes are picked once per node and axis on the packed words of all four children -- six selects --
    // instead of a min and a max per child and axis; the two planes of an axis are then one packed FMA.
    typedef R R2 __attribute__((ext_vector_type(2)));
    const bool nx = w.inv_d.x < R(0), ny = w.inv_d.y < R(0), nz = w.inv_d.z < R(0);
    const uint32_t qnx = nx ? q1.w : q1.x, qfx = nx ? q1.x : q1.w;
    const uint32_t qny = ny ? q2.x : q1.y, qfy = ny ? q1.y : q2.x;
    const uint32_t qnz = nz ? q2.y : q1.z, qfz = nz ? q1.z : q2.y;
    const int link[4] = {(int)q2.z, (int)q2.w, (int)q3.x, (int)q3.y};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        R2 qx, qy, qz;  // (near, far)
        qx.x = (R)byte_to_float(qnx, k); qx.y = (R)byte_to_float(qfx, k);
        qy.x = (R)byte_to_float(qny, k); qy.y = (R)byte_to_float(qfy, k);
        qz.x = (R)byte_to_float(qnz, k); qz.y = (R)byte_to_float(qfz, k);
        const R2 tx = qx * ax_ + bx_, ty = qy * ay_ + by_, tz = qz * az_ + bz_;
        const R en = fmax(fmax((R)tx.x, (R)ty.x), (R)tz.x);
        const R ex = fmin(fmin((R)tx.y, (R)ty.y), (R)tz.y) * widen;
        child[k] = link[k];
        const bool hit = en <= fmin(ex, t_limit) && ex >= R(0) && child[k] != BVH_NO_CHILD;
        dist[k] = hit ? en : inf;
    }
}

// Nearest of the children that were hit next, the others on the stack with the farthest deepest.
template <typename R, typename Push, typename Pop>
RRT_DEV void node_descend(TreeWalk<R>& w, R (&dist)[4], int (&child)[4], Push push, Pop pop) {
    const R inf = Num<R>::tmax();
    // sorting network for four keys
    order2(dist[0], child[0], dist[1], child[1]);
    order2(dist[2], child[2], dist[3], child[3]);
    order2(dist[0], child[0], dist[2], child[2]);
    order2(dist[1], child[1], dist[3], child[3]);
    order2(dist[1], child[1], dist[2], child[2]);
    if (dist[3] < inf) push(child[3]);
    if (dist[2] < inf) push(child[2]);
    if (dist[1] < inf) push(child[1]);
    if (dist[0] < inf) w.node = child[0];
    else pop();
}

template <typename R>
RRT_DEV void walk_node(TreeWalk<R>& w, TreeStackLds& lds, TreeStackSpill& spill, uint4 q0, uint4 q1, uint4 q2, uint4 q3,
                       unsigned long long* overflow, R t_limit) {
    R dist[4];
    int child[4];
    node_children(w, q0, q1, q2, q3, dist, child, t_limit);
    node_descend(w, dist, child, [&](int c) { walk_push(w, lds, spill, c, overflow); }, [&]() { walk_pop(w, lds, spill); });
}

// One leaf primitive, given as the four 16-byte words of its LeafRec<float> record, against the ray.
// Triangles and parallelograms are tested from these words with the arithmetic of prim_test_as, in
// the same order (the scan and the tree must agree bit for bit); other types from their PrimRec.
template <typename R>
RRT_DEV void leaf_test_words(const SceneView<R>& sv, uint4 q0, uint4 q1, uint4 q2, uint4 q3, V3<R> o, V3<R> d,
                             HitAcc<R>& acc) {
#pragma clang fp contract(off)
    static_assert(sizeof(R) == 4, "leaf words are an f32 record");
    const int type = (int)q3.x, g = (int)q3.y;
    R t;
    bool ok;
    if (type == PRIM_TRIANGLE || type == PRIM_PGRAM) {
        const R f0 = __uint_as_float(q0.x), f1 = __uint_as_float(q0.y), f2 = __uint_as_float(q0.z), f3 = __uint_as_float(q0.w);
        const R f4 = __uint_as_float(q1.x), f5 = __uint_as_float(q1.y), f6 = __uint_as_float(q1.z), f7 = __uint_as_float(q1.w);
        const R f8 = __uint_as_float(q2.x), f9 = __uint_as_float(q2.y), f10 = __uint_as_float(q2.z), f11 = __uint_as_float(q2.w);
        const R best_t = Num<R>::tmax();
        R nd, no;
        plane_nd_no<R, R>(f0, f1, f2, f3, o, d, nd, no);
        t = no * Num<R>::rcp(nd);
        R px, py, pz;
        plane_point<R, R>(t, o, d, px, py, pz);
        R a = plane_coeff<R>(f4, f5, f6, f7, px, py, pz);
        R b = plane_coeff<R>(f8, f9, f10, f11, px, py, pz);
        R det_ok = Num<R>::abs(nd) - Num<R>::eps();
        if (type == PRIM_PGRAM) {
            R h = min3(R(0.5) - Num<R>::abs(a), R(0.5) - Num<R>::abs(b), det_ok);
            ok = min3(h, t - Num<R>::tmin(), headroom(best_t, t)) >= R(0);
        } else {
            R h = min3(min3(a, b, R(1) - a), R(1) - (a + b), det_ok);
            ok = (h >= R(0)) && (fmin(t - Num<R>::tmin(), headroom(best_t, t)) > R(0));
        }
    } else {
        ok = prim_test(as_constant(sv.prims) + g, o, d, Num<R>::tmax(), t);  // sphere, general quad
    }
    if (ok) hit_consider(acc, t, g, type != PRIM_TRIANGLE);
}

// One leaf primitive read through its record (any precision).
template <typename R, bool TRACK = false>
RRT_DEV void leaf_test(const SceneView<R>& sv, const LeafRec<R>* leaf, V3<R> o, V3<R> d, HitAcc<R>& acc, HitTrack<TRACK>* tr = nullptr) {
    const int type = leaf->type;
    const int g = leaf->g;
    const PrimRec<R> RRT_CONSTANT* as_prim = (const PrimRec<R> RRT_CONSTANT*)(const void*)leaf;
    const ErrRec RRT_CONSTANT* er = nullptr;
    if constexpr (TRACK) er = as_constant(tr->err) + g;
    R t;
    bool ok;
    if (type == PRIM_TRIANGLE) {
        ok = prim_test_as<R, PRIM_TRIANGLE, TRACK>(as_prim, o, d, Num<R>::tmax(), t, tr, er);  // reads f[0..11] only
    } else if (type == PRIM_PGRAM) {
        ok = prim_test_as<R, PRIM_PGRAM, TRACK>(as_prim, o, d, Num<R>::tmax(), t, tr, er);     // reads f[0..11] only
    } else {
        ok = prim_test<R, TRACK>(as_constant(sv.prims) + g, o, d, Num<R>::tmax(), t, tr, er);   // sphere, general quad
    }
    if (ok) hit_consider<R, TRACK>(acc, t, g, type != PRIM_TRIANGLE, tr);
}

// The primitives kept out of the tree (unbounded planes, bent quads), scanned wave-uniformly.
template <typename R, bool TRACK = false>
RRT_DEV void scan_beside_tree(const SceneView<R>& sv, V3<R> o, V3<R> d, HitAcc<R>& acc, HitCounters& hc, HitTrack<TRACK>* tr = nullptr) {
    const PrimRec<R> RRT_CONSTANT* prims = as_constant(sv.prims);
    const int32_t RRT_CONSTANT* lin = as_constant(sv.linear_prim);
    for (int k = 0; k < sv.n_linear_prims; ++k) {
        const int g = lin[k];
        R t;
        const ErrRec RRT_CONSTANT* er = nullptr;
        if constexpr (TRACK) er = as_constant(tr->err) + g;
        if (prim_test<R, TRACK>(prims + g, o, d, Num<R>::tmax(), t, tr, er))
            hit_consider<R, TRACK>(acc, t, g, prims[g].type != PRIM_TRIANGLE, tr);
    }
    hc.prim_tests += (uint32_t)sv.n_linear_prims;
}

// Scene::intersect through the tree (plus a scan of the primitives kept out of it): one ray per
// lane from start to end (parity kernels and the fused renderer; the wavefront pipeline has a
// persistent variant of this loop).
template <typename R, bool TRACK = false>
RRT_DEV int closest_hit_bvh(const SceneView<R>& sv, V3<R> o, V3<R> d, R& t_out, HitCounters& hc, HitTrack<TRACK>* tr = nullptr) {
    __shared__ TreeStackLds s_stack;
    TreeStackSpill spill;
    TreeWalk<R> w;
    hit_init(w.acc);
    scan_beside_tree<R, TRACK>(sv, o, d, w.acc, hc, tr);
    if (sv.n_bvh_prims == 1) {
        leaf_test<R, TRACK>(sv, sv.leaves, o, d, w.acc, tr);
        hc.prim_tests += 1;
    } else if (sv.n_bvh_prims > 1) {
        walk_start(w, o, d);
        const int top_n = tree_top_count(sv);
        while (w.node != BVH_DONE) {
            if (w.node >= 0) {
                uint4 q0, q1, q2, q3;
                tree_node_words(sv, w.node, top_n, q0, q1, q2, q3);
                R t_limit = w.acc.t;
                if constexpr (TRACK) t_limit += 2.f * tr->best_err;
                walk_node(w, s_stack, spill, q0, q1, q2, q3, sv.stack_overflow, t_limit);
                hc.node_tests += 1;
            }
            if (w.node < 0 && w.node != BVH_DONE) {  // (in the same trip as the node step that led here)
                const int v = ~w.node;
                const int slot = v & BVH_SLOT_MASK, left = v >> BVH_SLOT_BITS;
                leaf_test<R, TRACK>(sv, sv.leaves + slot, w.o, w.d, w.acc, tr);
                hc.prim_tests += 1;
                if (left > 0) w.node = ~((slot + 1) | ((left - 1) << BVH_SLOT_BITS));
                else walk_pop(w, s_stack, spill);
            }
        }
    }
    t_out = w.acc.t;
    return hit_winner(w.acc);
}

// Closest hit over the whole scene.  Returns the scan-order primitive index or -1.
template <typename R, bool BVH, bool TRACK = false>
RRT_DEV int closest_hit(const SceneView<R>& sv, V3<R> o, V3<R> d, R& t, HitCounters& hc, HitTrack<TRACK>* tr = nullptr) {
    if (BVH) return closest_hit_bvh<R, TRACK>(sv, o, d, t, hc, tr);
    R best_t = Num<R>::tmax();
    int best_g = -1;
    scan_linear<R, TRACK>(sv, o, d, best_t, best_g, hc, tr);
    t = best_t;
    return best_g;
}

// The closest hit of the f32 kernels, made to name the primitive the f64 arithmetic names: the f32 query is tracked
// (HitTrack), and a lane whose query met a decision inside its error bound repeats it in f64 on the f64 tables, with the
// ray (o64, d64) -- the f64 ray the f32 one was rounded from (e_in = 1), or the f32 ray itself widened (e_in = 0).
// The f32 distance stands unless f64 names another primitive.  The linear scan reads its tables wave-uniformly, so all
// lanes of a wave run the f64 scan when one of them needs it; a walk through the tree is per lane.
// Returns the primitive; `redone` says whether this lane asked for f64.
template <bool BVH>
RRT_DEV int closest_hit_exact(const SceneView<float>& sv, const SceneView<double>& sv64, const ErrRec* err, V3<float> o, V3<float> d,
                              V3<double> o64, V3<double> d64, float e_in, float& t, HitCounters& hc, bool& redone) {
    HitTrack<true> tr;
    track_start(tr, err, o, e_in);
    int g = closest_hit<float, BVH, true>(sv, o, d, t, hc, &tr);
    redone = tr.amb;
    if (__ballot(tr.amb) != 0ull) {
        if (!BVH || tr.amb) {
            double t64;
            HitCounters hc64{0, 0};
            const int g64 = closest_hit<double, BVH>(sv64, o64, d64, t64, hc64);
            if (tr.amb && g64 != g) {
                g = g64;
                t = g64 >= 0 ? (float)t64 : Num<float>::tmax();
            }
        }
    }
    return g;
}

}  // namespace rrt
